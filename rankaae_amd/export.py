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
