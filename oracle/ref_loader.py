"""
TEST INFRASTRUCTURE — never imported by the product path.

File-level loader for the *reference* hot-path modules (read-only mount at
/root/reference).  It exists only so that `oracle/gen_golden.py` can run the
reference's own Python on seeded inputs IN THIS CONTAINER and write golden
input/output vectors to `tests/golden/`.  Nothing of the reference (source,
bytecode, pickles) is copied into this repository and nothing here runs on the
GPU box (where /root/reference does not exist).

Why a loader: `import nicr_mt_scene_analysis` raises an ordinary
ModuleNotFoundError (`nicr_scene_analysis_datasets`, `cv2`, `torchmetrics`,
`termcolor` are not installed; no network).  The hot-path modules themselves
only need torch/numpy/scipy, so each file is loaded individually with
`importlib.util.spec_from_file_location` under empty stub parent packages.
Two third-party stand-ins are provided (they are NOT reference code):
  * `cv2`          : empty module (only touched inside `resize()` bodies that
                     the hot path never reaches),
  * `torchmetrics` : a 15-line `Metric` (nn.Module + add_state/reset), the only
                     thing `metric/miou.py` and `metric/pq.py` use from it.
"""
import importlib.util
import os
import sys
import types

import torch

REF_ROOT = '/root/reference/src/nicr_mt_scene_analysis'
PKG = 'nicr_mt_scene_analysis'


def reference_available() -> bool:
    return os.path.isdir(REF_ROOT)


def _stub_package(name: str, path: str) -> types.ModuleType:
    if name in sys.modules:
        return sys.modules[name]
    mod = types.ModuleType(name)
    mod.__path__ = [path]
    mod.__package__ = name
    sys.modules[name] = mod
    return mod


def _load(modname: str, relpath: str) -> types.ModuleType:
    full = f'{PKG}.{modname}'
    if full in sys.modules:
        return sys.modules[full]
    spec = importlib.util.spec_from_file_location(
        full, os.path.join(REF_ROOT, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[full] = mod
    spec.loader.exec_module(mod)
    # make it reachable as attribute of its parent stub
    parent_name, _, child = full.rpartition('.')
    setattr(sys.modules[parent_name], child, mod)
    return mod


def _install_third_party_standins() -> None:
    if 'cv2' not in sys.modules:
        sys.modules['cv2'] = types.ModuleType('cv2')

    if 'torchmetrics' not in sys.modules:
        tm = types.ModuleType('torchmetrics')

        class Metric(torch.nn.Module):
            """Minimal state holder (add_state / reset), nothing else."""

            def __init__(self, **kwargs):
                super().__init__()
                self._defaults = {}

            def add_state(self, name, default, dist_reduce_fx=None):
                self._defaults[name] = default
                setattr(self, name, default.clone())

            def reset(self):
                for k, v in self._defaults.items():
                    setattr(self, k, v.clone())

        tm.Metric = Metric
        sys.modules['torchmetrics'] = tm

    if 'nicr_scene_analysis_datasets' not in sys.modules:
        # only `OrientationDict` (a typing alias) is imported by metric/mae.py
        ds = types.ModuleType('nicr_scene_analysis_datasets')
        ds.__path__ = []
        base = types.ModuleType('nicr_scene_analysis_datasets.dataset_base')
        base.OrientationDict = dict
        ds.dataset_base = base
        sys.modules['nicr_scene_analysis_datasets'] = ds
        sys.modules['nicr_scene_analysis_datasets.dataset_base'] = base


_loaded = None


def load_reference():
    """Returns a namespace with the reference hot-path symbols."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError('/root/reference is not mounted here')

    _install_third_party_standins()

    _stub_package(PKG, REF_ROOT)
    for sub in ('utils', 'loss', 'metric', 'data', 'data.preprocessing',
                'model', 'model.postprocessing'):
        _stub_package(f'{PKG}.{sub}',
                      os.path.join(REF_ROOT, sub.replace('.', '/')))
        parent, _, child = f'{PKG}.{sub}'.rpartition('.')
        setattr(sys.modules[parent], child, sys.modules[f'{PKG}.{sub}'])

    ns = types.SimpleNamespace()

    # --- utils (only the files the hot path touches) ---------------------------
    _load('utils._misc', 'utils/_misc.py')
    u_torch = _load('utils._torch', 'utils/_torch.py')
    u_ori = _load('utils._orientation', 'utils/_orientation.py')
    utils_pkg = sys.modules[f'{PKG}.utils']
    utils_pkg.biternion2rad = u_ori.biternion2rad
    utils_pkg.mps_cpu_fallback = u_torch.mps_cpu_fallback
    utils_pkg.to_cpu_if_mps_tensor = u_torch.to_cpu_if_mps_tensor
    utils_pkg.partial_class = sys.modules[f'{PKG}.utils._misc'].partial_class
    ns.panoptic_merge = _load('utils.panoptic_merge', 'utils/panoptic_merge.py')
    ns.biternion2rad = u_ori.biternion2rad

    # --- types / data helpers ------------------------------------------------
    _load('types', 'types.py')
    _load('data._types', 'data/_types.py')
    _load('data.preprocessing.base', 'data/preprocessing/base.py')
    _load('data.preprocessing.clone', 'data/preprocessing/clone.py')
    _load('data.preprocessing.utils', 'data/preprocessing/utils.py')
    ns.resize = _load('data.preprocessing.resize', 'data/preprocessing/resize.py')
    ns.APPLIED_PREPROCESSING_KEY = \
        sys.modules[f'{PKG}.data.preprocessing.base'].APPLIED_PREPROCESSING_KEY

    # --- target generators --------------------------------------
    sys.modules[f'{PKG}.data'].CollateIgnoredDict = \
        sys.modules[f'{PKG}.data._types'].CollateIgnoredDict
    ns.prep_instance = _load('data.preprocessing.instance', 'data/preprocessing/instance.py')
    ns.prep_panoptic = _load('data.preprocessing.panoptic', 'data/preprocessing/panoptic.py')
    ns.prep_dve = _load('data.preprocessing.dense_visual_embedding',
                        'data/preprocessing/dense_visual_embedding.py')

    # --- postprocessing ------------------------------------------------------
    _load('model.postprocessing.base', 'model/postprocessing/base.py')
    _load('model.postprocessing.dense_base', 'model/postprocessing/dense_base.py')
    ns.post_semantic = _load('model.postprocessing.semantic',
                             'model/postprocessing/semantic.py')
    ns.post_instance = _load('model.postprocessing.instance',
                             'model/postprocessing/instance.py')
    ns.post_panoptic = _load('model.postprocessing.panoptic',
                             'model/postprocessing/panoptic.py')

    # --- losses --------------------------------------------------------------
    _load('loss.base', 'loss/base.py')
    ns.loss_ce = _load('loss.ce', 'loss/ce.py')
    ns.loss_mse = _load('loss.mse', 'loss/mse.py')
    ns.loss_l1 = _load('loss.l1', 'loss/l1.py')
    ns.loss_vonmises = _load('loss.vonmises', 'loss/vonmises.py')
    ns.loss_cos_emb = _load('loss.cos_emb', 'loss/cos_emb.py')

    # --- metrics -------------------------------------------------------------
    ns.metric_miou = _load('metric.miou', 'metric/miou.py')
    ns.metric_pq = _load('metric.pq', 'metric/pq.py')
    ns.metric_mae = _load('metric.mae', 'metric/mae.py')

    _loaded = ns
    return ns


if __name__ == '__main__':
    ref = load_reference()
    print('loaded:', [k for k in vars(ref)])
