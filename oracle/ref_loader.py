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
For the task helpers (`load_reference(task_helpers=True)`) two more things are replaced by
no-ops, neither of which is on the path under test:
  * the reference's `visualization` sub-package (PIL / matplotlib colourisers the helpers call
    for the FIRST validation batch to build example images; needs matplotlib + fonts):
    functions that return None,
  * `torch.multiprocessing`'s spawn pool inside `metric/pq.py` (its workers would have to
    re-import the module by name, which only exists in this process): an in-process
    executor, so `PanopticQuality.update` runs the reference's own `compare_and_accumulate`
    synchronously and adds the per-image vectors in image order exactly as pq.py:291-296.
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
        base.SemanticLabelList = list
        ds.dataset_base = base
        sys.modules['nicr_scene_analysis_datasets'] = ds
        sys.modules['nicr_scene_analysis_datasets.dataset_base'] = base


_loaded = None


class _InlinePool:
    """stand-in for the spawn pool of metric/pq.py:211-218: same call surface, runs in-process"""

    class _Done:
        def __init__(self, value):
            self._value = value

        def get(self):
            return self._value

    def __init__(self, processes=None):
        pass

    def apply_async(self, fn, args=()):
        return self._Done(fn(*args))

    def terminate(self):
        pass

    close = join = terminate


def _load_task_helpers(ns):
    """task_helper/{base,semantic,instance,panoptic,dense_visual_embedding}.py, unmodified"""
    # `..loss` / `..metric` package attributes the helpers import by name
    loss_pkg, metric_pkg = sys.modules[f'{PKG}.loss'], sys.modules[f'{PKG}.metric']
    loss_pkg.CrossEntropyLossSemantic = ns.loss_ce.CrossEntropyLossSemantic
    loss_pkg.MSELoss = ns.loss_mse.MSELoss
    loss_pkg.L1Loss = ns.loss_l1.L1Loss
    loss_pkg.VonMisesLossBiternion = ns.loss_vonmises.VonMisesLossBiternion
    loss_pkg.CosineEmbeddingLoss = ns.loss_cos_emb.CosineEmbeddingLoss
    metric_pkg.MeanIntersectionOverUnion = ns.metric_miou.MeanIntersectionOverUnion
    metric_pkg.PanopticQuality = ns.metric_pq.PanopticQuality
    metric_pkg.PanopticQualityWithOrientationMAE = ns.metric_mae.PanopticQualityWithOrientationMAE
    # in-process executor instead of the spawn pool (see the module docstring)
    ns.metric_pq.mp = types.SimpleNamespace(
        cpu_count=lambda: 1,
        get_context=lambda method: types.SimpleNamespace(Pool=_InlinePool))
    # no-op visualisation (example images of the first validation batch only)
    vis = types.ModuleType(f'{PKG}.visualization')

    class PanopticColorGenerator:
        def __init__(self, *args, **kwargs):
            pass

    vis.PanopticColorGenerator = PanopticColorGenerator
    for name in ('visualize_semantic_pil', 'visualize_heatmap_pil', 'visualize_instance_center_pil',
                 'visualize_instance_offset_pil', 'visualize_instance_pil',
                 'visualize_instance_orientations_pil', 'visualize_orientation_pil',
                 'visualize_panoptic_pil'):
        setattr(vis, name, lambda *a, **k: None)
    sys.modules[f'{PKG}.visualization'] = vis
    setattr(sys.modules[PKG], 'visualization', vis)

    utils_pkg = sys.modules[f'{PKG}.utils']
    u_ori = sys.modules[f'{PKG}.utils._orientation']
    utils_pkg.np_rad2biternion = u_ori.np_rad2biternion
    utils_pkg.np_biternion2rad = getattr(u_ori, 'np_biternion2rad', None)
    _load('data.preprocessing.multiscale_supervision', 'data/preprocessing/multiscale_supervision.py')
    ns.prep_orientation = _load('data.preprocessing.orientation', 'data/preprocessing/orientation.py')
    _stub_package(f'{PKG}.task_helper', os.path.join(REF_ROOT, 'task_helper'))
    setattr(sys.modules[PKG], 'task_helper', sys.modules[f'{PKG}.task_helper'])
    _load('task_helper.base', 'task_helper/base.py')
    ns.th_semantic = _load('task_helper.semantic', 'task_helper/semantic.py')
    ns.th_instance = _load('task_helper.instance', 'task_helper/instance.py')
    ns.th_panoptic = _load('task_helper.panoptic', 'task_helper/panoptic.py')
    ns.th_dve = _load('task_helper.dense_visual_embedding', 'task_helper/dense_visual_embedding.py')


def load_reference(task_helpers: bool = False):
    """Returns a namespace with the reference hot-path symbols."""
    global _loaded
    if _loaded is not None:
        if task_helpers and not hasattr(_loaded, 'th_semantic'):
            _load_task_helpers(_loaded)
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
    if task_helpers:
        _load_task_helpers(ns)
    return ns


if __name__ == '__main__':
    ref = load_reference()
    print('loaded:', [k for k in vars(ref)])
