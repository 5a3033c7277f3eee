"""
`PanopticPostprocessing` on the MI355X
(reference model/postprocessing/panoptic.py:23-316).

The reference runs softmax -> max -> isin -> per-image nearest-center loops ->
`.cpu()` -> per-instance Python loops of `deeplab_merge_batch`.  Here the
inference path is four stream-ordered HIP launches (ops.panoptic_pipeline):
center NMS/top-k, ONE fused pass over the logits (argmax + foreground +
offset grouping + per-instance class votes), the per-instance class/rank
assignment and the paint kernel; everything stays on the GPU and the Python
dicts (`..._ids`, `..._instance_meta`) are built from one small D2H copy.
"""
import os
from typing import Dict, List, Tuple

import numpy as np
import torch

from ... import ops
from ...data.preprocessing.resize import get_fullres_key
from ...data.preprocessing.resize import get_valid_region_slices_and_fullres_shape
from ...types import BatchType
from ...types import DecoderRawOutputType
from ...types import PostprocessingOutputType
from ._lazy import LazyDict
from .dense_base import DensePostprocessingBase
from .instance import InstancePostprocessing
from .semantic import SemanticPostprocessing


class _HostTables:
    """the per-image tables of one pipeline run on the host, fetched (or awaited) on first use"""

    def __init__(self, fetch) -> None:
        self._fetch = fetch
        self._host = None

    def get(self) -> dict:
        if self._host is None:
            self._host = self._fetch()
            self._fetch = None
        return self._host


_PACK_VIA_DEVICE = os.environ.get('NMSA_PACK_VIA_DEVICE', '0') == '1'

class PanopticPostprocessing(DensePostprocessingBase):
    def __init__(
        self,
        semantic_postprocessing: SemanticPostprocessing,
        instance_postprocessing: InstancePostprocessing,
        semantic_classes_is_thing: Tuple[bool],
        semantic_class_has_orientation: Tuple[bool],
        normalized_offset: bool = True,
        compute_scores: bool = False,
        defer_host_sync: bool = False,
        **kwargs
    ) -> None:
        super().__init__()
        self._semantic_postprocessing = semantic_postprocessing
        self._instance_postprocessing = instance_postprocessing

        self._is_thing = np.asarray(semantic_classes_is_thing, dtype=bool)
        self._has_orientation = np.asarray(semantic_class_has_orientation, dtype=bool)
        # class indices (network output domain, no void) / panoptic class values (+1)
        self._thing_class_ids = np.where(self._is_thing)[0]
        self._thing_ids_panoptic = self._thing_class_ids + 1
        self._orientation_ids = np.where(self._has_orientation)[0] + 1

        self._normalized_offset = normalized_offset
        self._compute_scores = compute_scores
        # Host synchronisation.  The only thing `postprocess` must know before it returns is
        # whether more than `_max_centers` tied centers survived the top-k (then it re-runs with a
        # larger table): the center COUNTS travel on a copy stream right behind the selection
        # kernel (tens of microseconds into the call) and are awaited while the streaming kernels
        # run, so the call returns with the GPU still busy and consecutive calls pipeline.  The
        # per-image tables behind the Python objects (id dicts, instance meta, orientation dicts)
        # travel with an asynchronous copy and are awaited when first read.
        # extension `defer_host_sync=True`: not even the counts are awaited — an overflow then
        # surfaces as a RuntimeError at the first read of a host-built entry.
        self._defer_host_sync = defer_host_sync
        self._count_slots: Dict[tuple, dict] = {}
        self._pinned_ring: Dict[tuple, list] = {}
        self._pinned_next: Dict[tuple, int] = {}
        self._max_instances_per_category = 1 << 16
        self._host_columns = 32          # table columns fetched per image (grows on demand)
        self._device_luts: Dict[torch.device, Tuple[torch.Tensor, torch.Tensor]] = {}

    @property
    def max_instances_per_category(self):
        return self._max_instances_per_category

    def _luts(self, device):
        if device not in self._device_luts:
            thing = torch.from_numpy(self._is_thing.astype(np.uint8)).to(device)
            # orientation lut in the panoptic class-value domain (0 = void)
            ori = np.zeros((len(self._has_orientation) + 1,), np.uint8)
            ori[1:] = self._has_orientation
            self._device_luts[device] = (thing, torch.from_numpy(ori).to(device))
        return self._device_luts[device]

    def _postprocess_training(
        self, data: DecoderRawOutputType, batch: BatchType
    ) -> PostprocessingOutputType:
        (s_output, i_output), (s_side_outputs, i_side_outputs) = data
        r = self._semantic_postprocessing._postprocess_training((s_output, s_side_outputs), batch)
        r.update(self._instance_postprocessing._postprocess_training((i_output, i_side_outputs), batch))
        return r

    def _postprocess_inference(
        self, data: DecoderRawOutputType, batch: BatchType
    ) -> PostprocessingOutputType:
        (s_output, i_output), (s_side_outputs, i_side_outputs) = data
        post = self._instance_postprocessing
        with_orientation = (len(i_output) == 3)
        if with_orientation:
            center_heatmap, center_offset, orientation = i_output
        else:
            center_heatmap, center_offset = i_output
        dev = s_output.device
        thing_lut, ori_lut = self._luts(dev)

        # ---- the hot path: 5 launches; ONE device->host copy (= the one sync) of the small
        #      per-image tables, cut to the number of columns recent batches needed -----------
        def run(on_centers=None):
            return ops.panoptic_pipeline(
                s_output, center_heatmap, center_offset, thing_lut,
                threshold=post._heatmap_threshold,
                kernel_size=post._heatmap_nms_kernel_size,
                top_k=post._top_k_instances,
                apply_foreground_mask=post._heatmap_apply_foreground_mask,
                normalized_offset=self._normalized_offset,
                distance_threshold=post._offset_distance_threshold,
                max_instances_per_category=self._max_instances_per_category,
                void_label=0, max_centers=post._max_centers,
                want_score=self._compute_scores, want_foreground=False,
                want_panoptic_semantic=False, on_centers=on_centers)
        if self._defer_host_sync:
            p = run()
            tables = _HostTables(self._fetch_tables_async(p, post, checked=False))
        else:
            while True:
                peek = {}
                p = run(on_centers=lambda cen: peek.update(self._peek_center_counts(cen)))
                peek['done'].synchronize()             # long there: the streaming kernels still run
                n_max = int(peek['host'].max()) if peek['host'].numel() else 0
                if n_max > post._max_centers:          # center table overflow: re-run, larger
                    post._max_centers = 1 << (n_max - 1).bit_length()
                    continue
                if n_max > self._host_columns:         # more instances than table columns fetched so far
                    self._host_columns = 1 << (n_max - 1).bit_length()
                break
            tables = _HostTables(self._fetch_tables_async(p, post, checked=True))

        # ---- semantic entries (semantic.py:46-80) -------------------------------------
        r = LazyDict(semantic_output=s_output, semantic_side_outputs=s_side_outputs)
        sem_u8 = p['semantic_idx_u8']
        r.set_lazy('semantic_softmax_scores', lambda: ops.semantic_softmax(s_output))
        if self._compute_scores:                  # the score maps need it: fused pass computes it
            r['semantic_segmentation_score'] = p['semantic_score']
        else:                                     # otherwise only when somebody reads it
            r.set_lazy('semantic_segmentation_score',
                       lambda: ops.semantic_argmax(s_output, want_u8=False, want_i64=False,
                                                   want_score=True)['score'])
        r.set_lazy('semantic_segmentation_idx', lambda: sem_u8.long())
        r.aux['semantic_segmentation_idx'] = sem_u8
        self._semantic_postprocessing._fullres_entries(r, s_output, batch)

        # ---- instance entries (instance.py:337-468): GT-foreground variants etc. ------
        r.merge(post._postprocess_inference((i_output, i_side_outputs), batch))

        # ---- panoptic entries (panoptic.py:118-167) ---------------------------------------
        panoptic_seg = p['panoptic']
        instance_seg = p['instance']
        # foreground = "the pixel's class is a thing" (panoptic.py:123-127): a lookup of the class map,
        # built when somebody reads it — the fused pass does not store a third per-pixel map
        # (small stores are what a read-dominated stream pays most for: DESIGN.md 5)
        r.set_lazy('panoptic_foreground_mask', lambda: thing_lut[sem_u8.long()].to(torch.bool))
        r['panoptic_segmentation_deeplab'] = panoptic_seg
        # what the map was painted from: the metric update reads these 2 B/px instead of the
        # 8 B/px map when the map it is handed IS this one (task_helper/panoptic.py)
        r.aux['panoptic_parts'] = {
            'panoptic': panoptic_seg, 'semantic_idx_u8': sem_u8, 'instance': instance_seg,
            'pan_of_inst': p['pan_of_inst'], 'is_thing': thing_lut, 'void_label': 0,
            'max_instances_per_category': self._max_instances_per_category}
        # id dicts / instance meta: Python objects built from the host tables when first read
        r.set_lazy('panoptic_segmentation_deeplab_ids',
                   lambda: self._id_dicts_from_host(tables.get()))
        # panoptic // max_instances (panoptic.py:160): 8 B/px more to write — built when read
        max_inst = self._max_instances_per_category
        r.set_derived('panoptic_segmentation_deeplab_semantic_idx',
                      lambda d: torch.div(d['panoptic_segmentation_deeplab'], max_inst,
                                          rounding_mode='floor'))
        r['panoptic_segmentation_deeplab_instance_idx'] = instance_seg
        ori_key = 'orientations_panoptic_segmentation_deeplab_instance'

        def _meta(d):
            h = tables.get()
            meta = InstancePostprocessing._meta_from_host(h['n_centers'], h['centers_yx'],
                                                          h['scores'], h['area'])
            if with_orientation:                  # panoptic.py:306-314
                for m, o in zip(meta, d[ori_key]):
                    for id_ in m:
                        m[id_]['orientation'] = o.get(id_, float('nan'))
            return meta
        r.set_derived('panoptic_segmentation_deeplab_instance_meta', _meta)
        if with_orientation:                      # panoptic.py:294-304, built when read
            def _orientation(d):
                fg = ori_lut[d['panoptic_segmentation_deeplab_semantic_idx']].to(torch.bool)
                return post._get_instance_orientation(orientation, instance_seg, fg)
            r.set_derived(ori_key, _orientation)

        if self._compute_scores:
            self._add_scores(r, p, r['panoptic_segmentation_deeplab_ids'],
                             r['panoptic_segmentation_deeplab_instance_meta'])

        # ---- full resolution (panoptic.py:242-291) --------------------------------------
        crop, shape = get_valid_region_slices_and_fullres_shape(batch, 'instance')

        def _fullres(t):
            return self._crop_to_valid_region_and_resize_prediction(
                t, valid_region_slices=crop, shape=shape, mode='nearest')

        keys = ['panoptic_segmentation_deeplab',
                'panoptic_segmentation_deeplab_instance_idx',
                'panoptic_segmentation_deeplab_semantic_idx']
        if self._compute_scores:
            keys += ['panoptic_segmentation_deeplab_semantic_score',
                     'panoptic_segmentation_deeplab_instance_score',
                     'panoptic_segmentation_deeplab_panoptic_score']
        for k in keys:
            if r.is_pending(k):
                r.set_derived(get_fullres_key(k), (lambda kk: (lambda d: _fullres(d[kk])))(k))
            else:
                r[get_fullres_key(k)] = _fullres(r[k])

        return r

    @staticmethod
    def _fetch_tables(p, columns: int) -> dict:
        """n_centers, centers, scores, areas, n_ids and the id tables of one pipeline run in ONE
        device->host copy (float64 holds every entry exactly: ids < 2^53, f32 scores)"""
        B, K = p['center_scores'].shape
        kc = max(1, min(int(columns), K))
        ki = min(kc, p['ids_pan'].shape[1])
        ka = min(kc + 1, p['area'].shape[1])
        flat_dev = torch.empty((B, 2 + 3 * kc + ka + 2 * ki), dtype=torch.float64,
                               device=p['center_scores'].device)
        L = ops.L
        L.check(L.lib().nmsa_pack_tables(
            L.ptr(p['n_centers']), L.ptr(p['n_ids']), L.ptr(p['centers_yx']),
            L.ptr(p['center_scores']), L.ptr(p['area']), L.ptr(p['ids_pan']), L.ptr(p['ids_ins']),
            B, K, kc, L.ptr(flat_dev), L.stream_ptr(flat_dev.device)), 'nmsa_pack_tables')
        return PanopticPostprocessing._split_tables(flat_dev.cpu().numpy(), B, kc, ka, ki)

    def _peek_center_counts(self, cen) -> dict:
        """asynchronous copy of the per-image center counts (B int32) on a copy stream, behind the
        selection kernel only -> {'host': pinned tensor, 'done': event}"""
        n = cen['n_centers']
        dev = n.device
        key = (dev, n.shape[0])
        slot = self._count_slots.get(key)
        if slot is None:
            slot = self._count_slots[key] = {
                'host': torch.empty(n.shape, dtype=n.dtype, pin_memory=True),
                'stream': torch.cuda.Stream(device=dev), 'ready': torch.cuda.Event(),
                'done': torch.cuda.Event()}
        cur = torch.cuda.current_stream(dev)
        slot['ready'].record(cur)
        with torch.cuda.stream(slot['stream']):
            slot['stream'].wait_event(slot['ready'])
            slot['host'].copy_(n, non_blocking=True)
            slot['done'].record(slot['stream'])
        n.record_stream(slot['stream'])
        return {'host': slot['host'], 'done': slot['done']}

    def _fetch_tables_async(self, p, post, checked: bool = False):
        """`defer_host_sync`: pack + asynchronous copy into pinned host memory now; the returned
        function waits for the copy (first read of a host-built entry) and parses it"""
        B, K = p['center_scores'].shape
        kc = max(1, min(int(self._host_columns), K))
        ki = min(kc, p['ids_pan'].shape[1])
        ka = min(kc + 1, p['area'].shape[1])
        dev = p['center_scores'].device
        shape = (B, 2 + 3 * kc + ka + 2 * ki)
        L = ops.L
        # pinned buffers come from a small ring owned by this object: allocating pinned memory per
        # call is slow, and the caching host allocator cannot recycle a block while the host runs
        # ahead of the GPU.  A slot that is reused while its previous result has not been read yet
        # first hands that result a private copy.
        slot = self._pinned_slot(shape)
        flat_host = slot['buffer']
        # The pack kernel stores the table STRAIGHT into the pinned host buffer (pinned memory is
        # mapped into the device's address space at the same address): no device staging tensor and
        # no copy kernel (5 us on the stream the host is waiting for).  NMSA_PACK_VIA_DEVICE=1: the
        # staged form.
        via_device = _PACK_VIA_DEVICE
        flat_dev = torch.empty(shape, dtype=torch.float64, device=dev) if via_device else flat_host
        L.check(L.lib().nmsa_pack_tables(
            L.ptr(p['n_centers']), L.ptr(p['n_ids']), L.ptr(p['centers_yx']),
            L.ptr(p['center_scores']), L.ptr(p['area']), L.ptr(p['ids_pan']), L.ptr(p['ids_ins']),
            B, K, kc, L.ptr(flat_dev), L.stream_ptr(dev)), 'nmsa_pack_tables')
        if via_device:
            flat_host.copy_(flat_dev, non_blocking=True)
        done = slot['event']
        done.record(torch.cuda.current_stream(dev))
        max_centers = post._max_centers
        state = {'flat': None}

        def detach():                           # called when the slot is about to be reused
            if state['flat'] is None:
                done.synchronize()
                state['flat'] = flat_host.numpy().copy()
        slot['detach'] = detach

        def finish():
            detach()
            if slot.get('detach') is detach:
                slot['detach'] = None
            host = self._split_tables(state['flat'], B, kc, ka, ki)
            n_max = max(host['n_centers'], default=0)
            if n_max > max_centers and not checked:    # (checked: the caller has seen the counts)
                post._max_centers = 1 << (n_max - 1).bit_length()
                raise RuntimeError(
                    f'{n_max} centers survived the top-k ties but the center table held '
                    f'{max_centers}: this result is truncated (defer_host_sync=True cannot re-run '
                    'it); the table has been enlarged for the following calls')
            if n_max > kc:                        # rare: more instances than columns fetched
                self._host_columns = 1 << (n_max - 1).bit_length()
                host = self._fetch_tables(p, self._host_columns)
            return host
        return finish

    def _pinned_slot(self, shape, ring: int = 8) -> dict:
        slots = self._pinned_ring.setdefault(shape, [])
        if len(slots) < ring:
            slots.append({'buffer': torch.empty(shape, dtype=torch.float64, pin_memory=True),
                          'event': torch.cuda.Event(), 'detach': None})
            self._pinned_next[shape] = 0
            return slots[-1]
        i = self._pinned_next[shape]
        self._pinned_next[shape] = (i + 1) % ring
        slot = slots[i]
        if slot['detach'] is not None:          # an unread result still points at this buffer
            slot['detach']()
            slot['detach'] = None
        else:
            slot['event'].synchronize()         # the previous copy into this buffer has landed
        return slot

    @staticmethod
    def _split_tables(flat, B, kc, ka, ki) -> dict:
        edges = np.cumsum([0, 1, 1, 2 * kc, kc, ka, ki, ki])
        cols = [flat[:, a:b] for a, b in zip(edges[:-1], edges[1:])]
        return {'columns': kc,
                'n_centers': cols[0][:, 0].astype(np.int64).tolist(),
                'n_ids': cols[1][:, 0].astype(np.int64).tolist(),
                'centers_yx': cols[2].reshape(B, kc, 2), 'scores': cols[3], 'area': cols[4],
                'ids_pan': cols[5], 'ids_ins': cols[6]}

    @staticmethod
    def _id_dicts_from_host(host) -> List[dict]:
        """{panoptic id: instance id} per image, insertion order = ascending instance id
        (panoptic_merge.py:195-213)"""
        pan = host['ids_pan'].astype(np.int64).tolist()
        ins = host['ids_ins'].astype(np.int64).tolist()
        return [dict(zip(pr[:n], ir[:n])) for pr, ir, n in zip(pan, ins, host['n_ids'])]

    # f3: score maps of panoptic.py:171-239 — two HIP kernels
    # (`nmsa_panoptic_scores`), no softmax tensor, no per-instance loops over images.
    def _add_scores(self, r, p, panoptic_ids: List[dict], meta: List[dict]) -> None:
        B = p['instance'].shape[0]
        dev = p['instance'].device
        score_tab = torch.zeros((B, 256), dtype=torch.float32, device=dev)
        k = min(p['center_scores'].shape[1], 255)
        score_tab[:, 1:k + 1] = p['center_scores'][:, :k]
        sc = ops.panoptic_scores(
            r['semantic_output'], p['semantic_idx_u8'], p['semantic_score'], p['instance'],
            p['panoptic'], p['pan_of_inst'], score_tab, self._max_instances_per_category)
        r['panoptic_segmentation_deeplab_semantic_score'] = sc['semantic_score']
        r['panoptic_segmentation_deeplab_instance_score'] = sc['instance_score']
        r['panoptic_segmentation_deeplab_panoptic_score'] = sc['panoptic_score']

        # ONE small D2H copy for the meta dicts (panoptic.py:214-232)
        host = torch.cat([sc['mean_semantic_score'].double(),
                          (p['pan_of_inst'] // self._max_instances_per_category).double()],
                         dim=1).cpu().tolist()
        for b in range(B):
            for pan_id, ins_id in panoptic_ids[b].items():
                m = meta[b][ins_id]
                mean = float(np.float32(host[b][ins_id]))
                m['semantic_score'] = mean
                m['semantic_idx'] = int(host[b][256 + ins_id])
                m['panoptic_score'] = float(np.float32(mean) * np.float32(m['score']))
                m['panoptic_id'] = pan_id
