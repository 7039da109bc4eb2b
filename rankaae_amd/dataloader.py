"""Dataset ingest with the reference's semantics (``sc/clustering/dataloader.py:8-77``):
CSV with two index columns, ``n_aux`` ``AUX_*`` columns then ``ENE_<eV>`` columns; contiguous
70/15/15 row split; float64 -> float32; training rows reshuffled every epoch; the last,
partial batch is kept.  Unlike the reference the CSV is parsed ONCE and the split arrays go
to the device whole -- batches are gathered there by index (``raae_gather_batch``)."""
import numpy as np
import torch


class AuxSpectraDataset:
    def __init__(self, spec, aux, grid=None, atom_index=None, metadata=None):
        self.spec, self.aux, self.grid = spec, aux, grid
        self.atom_index, self.metadata = atom_index, metadata

    def __len__(self):
        return self.spec.shape[0]

    def __getitem__(self, idx):
        aux = np.array([0.0]) if self.aux is None else self.aux[idx]
        return torch.Tensor(self.spec[idx]), torch.Tensor(aux)


class Loader:
    """Minimal stand-in for ``torch.utils.data.DataLoader``: what ``Trainer`` needs
    (``dataset``, ``batch_size``, ``len``) plus the reference's shuffle order."""

    def __init__(self, dataset, batch_size, shuffle):
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle

    def __len__(self):
        return -(-len(self.dataset) // self.batch_size)

    def epoch_permutation(self):
        """Row order of one epoch exactly as ``DataLoader(shuffle=True, num_workers=0)`` draws it
        (SURVEY.md finding 9): iterator ``_base_seed`` draw, sampler seed draw, private randperm."""
        torch.empty((), dtype=torch.int64).random_()
        if not self.shuffle:
            return torch.arange(len(self.dataset))
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        g = torch.Generator()
        g.manual_seed(seed)
        return torch.randperm(len(self.dataset), generator=g)

    def __iter__(self):
        perm = self.epoch_permutation()
        for i in range(len(self)):
            rows = perm[i * self.batch_size:(i + 1) * self.batch_size].numpy()
            aux = self.dataset.aux[rows] if self.dataset.aux is not None else np.zeros((len(rows), 1))
            yield torch.tensor(self.dataset.spec[rows], dtype=torch.float32), torch.tensor(aux, dtype=torch.float32)


def split_counts(n_rows, ratios=(0.7, 0.15, 0.15)):
    n = [int(n_rows * r) for r in ratios]
    n[-1] = int(n_rows) - sum(n[:-1])
    return n


def load_csv(csv_fn, n_aux):
    import pandas as pd
    df = pd.read_csv(csv_fn, index_col=[0, 1], comment="#")
    cols = df.columns.to_list()
    assert "ENE_" in cols[n_aux]
    if n_aux > 0:
        assert "ENE_" not in cols[n_aux - 1]
        assert "AUX_" in cols[0]
        assert "AUX_" in cols[n_aux - 1]
    grid = np.array([float(c.strip("ENE_")) for c in cols if c.startswith("ENE_")])
    data = df.to_numpy()
    return data[:, n_aux:], (data[:, :n_aux] if n_aux > 0 else None), grid, df.index.to_list()


def get_dataloaders(csv_fn, batch_size, train_val_test_ratios=(0.7, 0.15, 0.15), n_aux=0, arrays=None):
    """``arrays=(spec, aux)`` bypasses the CSV (synthetic data already in memory)."""
    if arrays is None:
        spec, aux, grid, index = load_csv(csv_fn, n_aux)
    else:
        spec, aux = arrays
        grid, index = None, list(range(len(spec)))
    n = split_counts(len(spec), train_val_test_ratios)
    loaders, lo = [], 0
    meta = {"path": csv_fn, "train_test_val_split_ratio": train_val_test_ratios}
    for i, cnt in enumerate(n):
        ds = AuxSpectraDataset(spec[lo:lo + cnt], None if aux is None else aux[lo:lo + cnt], grid,
                               index[lo:lo + cnt], meta)
        loaders.append(Loader(ds, batch_size, shuffle=(i == 0)))
        lo += cnt
    return loaders
