"""Latent-space export: the ``Reconstruct`` evaluator of the reference's report tool
(``sc/report/analysis_new.py:94-129``) on the HIP engine.

``evaluate(test_ds, trainer)`` runs the eval-mode encoder and decoder over ``test_ds.spec`` on the GPU and
``to_file(dir)`` writes ``<name>_spec_in.txt``, ``<name>_spec_out.txt`` and ``<name>_styles.txt`` with
``numpy.savetxt``'s defaults (``%.18e``, space separated) -- the files the reference writes and whose sample
(``sc/tests/data/recon_styles.txt``: one row per spectrum, ``nstyle`` columns) its tests keep.
"""
import os

import numpy as np
import torch


class Reconstruct:
    def __init__(self, device=None, name="reconstructed"):
        self.device, self.name = device, name
        self.result, self.metadata = {}, {}

    def evaluate(self, test_ds, model, path_to_save=None):
        """``test_ds``: anything with a ``.spec`` array ``[n, L]`` (``AuxSpectraDataset``); ``model``: a
        ``rankaae_amd.trainer.Trainer`` (its engine holds the trained weights) or a ``StepEngine``."""
        eng = getattr(model, "engine", model)
        if not hasattr(eng, "reconstruct"):
            raise TypeError("Reconstruct.evaluate needs a rankaae_amd Trainer or StepEngine: the export runs on the "
                            "HIP engine (a final.pt dict of plain modules can be evaluated with PyTorch directly)")
        spec_in = torch.as_tensor(np.asarray(test_ds.spec), dtype=torch.float32).contiguous().to(eng.device)
        styles, spec_out = eng.reconstruct(spec_in)
        self.metadata.update(name=self.name, data=getattr(test_ds, "metadata", {}).get("path"))
        self.result.update(input=spec_in.cpu().numpy(), styles=styles.cpu().numpy(), output=spec_out.cpu().numpy())
        if path_to_save is not None:
            self.to_file(path_to_save)
        return self.result

    def to_file(self, path_to_save):
        file_path = os.path.join(path_to_save, self.name)
        np.savetxt(file_path + "_spec_in" + ".txt", self.result["input"])
        np.savetxt(file_path + "_spec_out" + ".txt", self.result["output"])
        np.savetxt(file_path + "_styles" + ".txt", self.result["styles"])


def spectra_variation(model, istyle, styles, n_spec=50, n_sampling=1000):
    """The numbers behind the report's style-variation plots (``plot_spectra_variation``,
    sc/report/analysis.py:33-86): style ``istyle`` swept over the 5th..95th percentile of its column in
    ``styles`` in ``n_spec`` steps; with ``n_sampling == 0`` the other styles are 0, otherwise each point is the
    mean decoded spectrum over ``n_sampling`` draws of the other styles from N(0, 1).  The decoder and the
    averaging run on the HIP engine.  Returns ``(style_variation[n_spec], spec_out[n_spec, L])`` as numpy."""
    eng = getattr(model, "engine", model)
    styles = np.asarray(styles)
    assert styles.ndim == 2 and styles.shape[1] == eng.nstyle
    left, right = np.percentile(styles[:, istyle], [5, 95])
    if n_sampling == 0:
        c = np.linspace(left, right, n_spec)
        con_c = torch.zeros(n_spec, eng.nstyle)
        con_c[:, istyle] = torch.tensor(c, dtype=torch.float)
        return c, eng.decode(con_c.to(eng.device)).cpu().numpy()
    con_c = torch.randn([n_spec, n_sampling, eng.nstyle], device=eng.device)
    variation = torch.linspace(left, right, n_spec, device=eng.device)
    con_c[..., istyle] = variation[:, None]
    spec_out = eng.decode(con_c.reshape(n_spec * n_sampling, eng.nstyle), n_sampling=n_sampling)
    return variation.cpu().numpy(), spec_out.cpu().numpy()
