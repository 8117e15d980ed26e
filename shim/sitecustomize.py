"""Path-shadowing adoption (Option A): put this directory on PYTHONPATH when running the reference's own
scripts from a reference checkout,

    cd /path/to/reference && PYTHONPATH=/path/to/this/repo/shim python -m sample.demo_style_transfer ...

Python imports `sitecustomize` at start-up; it installs one import hook that answers the module names of the denoise path
with this repository's engine-backed counterparts (same classes, signatures, state-dict layout) and leaves every other
module of the reference's `diffusion/`, `model/`, `utils/`, `train/` packages (parser_util, dist_util, fixseed, nn, losses,
smpl, rotation2xyz, train_platforms, ...) to the checkout.  No file of the reference is edited."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# reference module (file:what the scripts take from it)                         -> counterpart
SHADOW = {
    "diffusion.gaussian_diffusion": "mst_amd.diffusion.gaussian_diffusion",                # utils/model_util.py:1
    "diffusion.respace": "mst_amd.diffusion.respace",                                      # utils/model_util.py:2
    "diffusion.inpainting_gaussian_diffusion": "mst_amd.diffusion.inpainting_gaussian_diffusion",   # demo:17, finetune:19
    "model.mdm_forstyledataset": "mst_amd.model.mdm_forstyledataset",                      # demo:57-62, finetune:75-91
    "model.cfg_sampler": "mst_amd.model.cfg_sampler",                                      # demo:159
    "utils.model_util": "mst_amd.utils.model_util",                                        # demo:14, finetune:16
    "train.training_loop": "mst_amd.train.training_loop",                                  # finetune:14
}


class _Alias(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        if REPO not in sys.path:
            sys.path.insert(0, REPO)
        import mst_amd  # noqa: F401  (loads the hyphenated package directory under an importable name)
        return importlib.import_module(self.target)

    def exec_module(self, module):
        pass


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname in SHADOW:
            return importlib.util.spec_from_loader(fullname, _Alias(SHADOW[fullname]))
        return None


if not any(isinstance(f, _Finder) for f in sys.meta_path):
    sys.meta_path.insert(0, _Finder())
