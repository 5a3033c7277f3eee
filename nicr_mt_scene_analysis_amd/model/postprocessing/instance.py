"""
`InstancePostprocessing` on the MI355X
(reference model/postprocessing/instance.py:24-468).

Same constructor kwargs, same private helpers and the same output-dict keys as
the reference; the per-pixel work runs in the HIP kernels of csrc/center_nms.hip
and csrc/panoptic.hip (center NMS + top-k + ordered compaction, exact
nearest-center grouping, per-instance biternion sums).  Each helper does at
most ONE small device->host copy (the per-image tables the Python-side dicts
are built from); the dense maps never leave the GPU.
"""
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from ... import ops
from ...data.preprocessing.resize import get_fullres_key
from ...data.preprocessing.resize import get_valid_region_slices_and_fullres_shape
from ...types import BatchType
from ...types import DecoderRawOutputType
from ...types import PostprocessingOutputType
from .dense_base import DensePostprocessingBase


class InstancePostprocessing(DensePostprocessingBase):
    def __init__(
        self,
        heatmap_threshold: float = 0.1,
        heatmap_nms_kernel_size: int = 3,
        heatmap_apply_foreground_mask: bool = False,
        top_k_instances: int = 64,
        normalized_offset: bool = True,
        offset_distance_threshold: Union[None, int] = None,
        **kwargs
    ) -> None:
        super().__init__()
        assert heatmap_nms_kernel_size % 2 == 1
        assert top_k_instances <= 254
        self._heatmap_threshold = heatmap_threshold
        self._heatmap_nms_kernel_size = heatmap_nms_kernel_size
        self._heatmap_nms_padding = (heatmap_nms_kernel_size - 1) // 2
        self._heatmap_apply_foreground_mask = heatmap_apply_foreground_mask
        self._top_k_instances = top_k_instances
        self._normalized_offset = normalized_offset
        self._offset_distance_threshold = offset_distance_threshold
        # capacity of the device-side center table; grows when ties at the k-th
        # value keep more than this many centers (the kernel reports the true count)
        self._max_centers = 256
        self.debug = kwargs.get('debug', False)

    # ------------------------------------------------------------------ centers
    def _run_center_kernel(self, center_heatmap, foreground_mask, want_mask=False):
        while True:
            cen = ops.center_nms_topk(
                center_heatmap, foreground_mask,
                threshold=self._heatmap_threshold,
                kernel_size=self._heatmap_nms_kernel_size,
                top_k=self._top_k_instances,
                apply_foreground_mask=self._heatmap_apply_foreground_mask,
                max_centers=self._max_centers, want_mask=want_mask)
            n_host = cen['n_centers'].cpu()
            n_max = int(n_host.max()) if n_host.numel() else 0
            if n_max <= self._max_centers:
                cen['n_host'] = n_host.tolist()
                return cen
            self._max_centers = 1 << (n_max - 1).bit_length()     # re-run, larger table

    def _get_instance_centers(
        self,
        center_heatmap: torch.Tensor,
        foreground_mask: Optional[torch.Tensor] = None,
    ) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        """-> (bool [B,H,W], list of int32 [n_b, 2] (y, x) in raster order)"""
        cen = self._run_center_kernel(center_heatmap, foreground_mask, want_mask=True)
        centers = [cen['centers_yx'][b, :n] for b, n in enumerate(cen['n_host'])]
        return cen['center_mask'], centers

    # ------------------------------------------------------------- segmentation
    @staticmethod
    def _meta_from_host(n_host, centers_yx, scores, area) -> List[Dict[int, dict]]:
        """host (numpy) tables -> the reference's per-image meta dicts (instance.py:255-266)"""
        K = scores.shape[1]
        nmax = max(1, min(max(n_host) if len(n_host) else 1, K))
        ka = min(nmax, 255)
        yx = centers_yx[:, :nmax].astype(np.int64).tolist()
        sc = scores[:, :nmax].astype(np.float64).tolist()
        ar = area[:, 1:ka + 1].astype(np.int64).tolist()
        meta = []
        for b, n in enumerate(n_host):
            n = min(n, nmax)
            # ids wrap at 256 (uint8, instance.py:236): bincount(minlength=n+1) has no entries
            # beyond 255
            areas = ar[b][:min(n, ka)] + [0] * max(0, n - ka)
            meta.append({
                i + 1: {'center_yx': (y, x), 'area': a, 'score': s_}
                for i, ((y, x), s_, a) in enumerate(zip(yx[b][:n], sc[b][:n], areas))})
        return meta

    def _segment(self, center_heatmap, center_offset, foreground_mask, scale_y, scale_x):
        """centers + grouping + meta with ONE device->host copy (= one sync) per call: the center
        list, scores and areas travel packed by `nmsa_pack_tables`; more tied centers than the
        device table holds -> larger table, run again"""
        while True:
            cen = ops.center_nms_topk(
                center_heatmap, foreground_mask,
                threshold=self._heatmap_threshold,
                kernel_size=self._heatmap_nms_kernel_size,
                top_k=self._top_k_instances,
                apply_foreground_mask=self._heatmap_apply_foreground_mask,
                max_centers=self._max_centers)
            grp = ops.group_offsets(center_offset, foreground_mask, cen['centers_yx'],
                                    cen['n_centers'], scale_y, scale_x,
                                    self._offset_distance_threshold)
            B, K = cen['scores'].shape
            ka = min(K + 1, 256)
            flat_dev = torch.empty((B, 2 + 3 * K + ka), dtype=torch.float64, device=cen['scores'].device)
            L = ops.L
            L.check(L.lib().nmsa_pack_tables(
                L.ptr(cen['n_centers']), None, L.ptr(cen['centers_yx']), L.ptr(cen['scores']),
                L.ptr(grp['area']), None, None, B, K, K, L.ptr(flat_dev),
                L.stream_ptr(flat_dev.device)), 'nmsa_pack_tables')
            flat = flat_dev.cpu().numpy()
            n_host = flat[:, 0].astype(np.int64).tolist()
            n_max = max(n_host, default=0)
            if n_max > self._max_centers:
                self._max_centers = 1 << (n_max - 1).bit_length()
                continue
            meta = self._meta_from_host(n_host, flat[:, 2:2 + 2 * K].reshape(B, K, 2),
                                        flat[:, 2 + 2 * K:2 + 3 * K], flat[:, 2 + 3 * K:])
            return grp['instance'], meta

    def _get_instance_segmentation(
        self,
        center_heatmap: torch.Tensor,
        center_offset: torch.Tensor,
        foreground_mask: torch.Tensor
    ) -> Tuple[torch.Tensor, List[Dict[int, dict]]]:
        """`center_offset` is expected de-normalised (pixels), as in the reference
        where the caller multiplies by (H, W) first (instance.py:361-365)."""
        return self._segment(center_heatmap, center_offset, foreground_mask, 1.0, 1.0)

    # -------------------------------------------------------------- orientation
    def _get_instance_orientation(
        self,
        orientation: torch.Tensor,
        instance_segmentation: torch.Tensor,
        foreground_mask: Optional[torch.Tensor]
    ) -> List[Dict[int, float]]:
        seg = instance_segmentation
        if seg.ndim == 4:
            seg = seg[:, 0]
        if seg.dtype == torch.bool:
            seg = seg.view(torch.uint8)
        if seg.dtype != torch.uint8:
            return self._get_instance_orientation_wide(orientation, seg, foreground_mask)
        r = ops.instance_orientation_sums(orientation, seg.contiguous(), foreground_mask)
        sums = r['sums'].to(torch.float32)
        # angle = atan2(sum(sin), sum(cos))  (utils/_orientation.py:39-42)
        angle = torch.atan2(sums[..., 1], sums[..., 0])
        packed = torch.cat([angle.to(torch.float64), r['count'].to(torch.float64)], dim=1).cpu().tolist()
        out = []
        for row in packed:
            out.append({i: row[i] for i in range(1, 256) if row[256 + i] > 0})
        return out

    def _get_instance_orientation_wide(self, orientation, seg, foreground_mask):
        """ground-truth instance maps (uint16 ids stored as int32 / int64, instance.py:432-438):
        ids are ranked on the device, the dict is keyed by the original ids"""
        max_instances = 1024
        while True:
            r = ops.instance_orientation_sums_wide(orientation, seg.contiguous(), foreground_mask,
                                                   max_instances)
            head = torch.cat([r['status'], r['n_ids']]).cpu().tolist()
            if head[0] & 1 and max_instances < 4096:
                max_instances = 4096
                continue
            break
        if head[0] & 32:
            raise NotImplementedError('instance ids outside [0, 65535] are not supported '
                                      '(dataset instance maps are uint16)')
        if head[0] & 1:
            raise NotImplementedError('more than 4096 distinct instance ids in one image')
        nmax = max(1, max(head[1:]))
        sums = r['sums'][:, :nmax].to(torch.float32)
        angle = torch.atan2(sums[..., 1], sums[..., 0])
        packed = torch.cat([r['ids'][:, :nmax].to(torch.float64), angle.to(torch.float64),
                            r['count'][:, :nmax].to(torch.float64)], dim=1).cpu().tolist()
        out = []
        for b, row in enumerate(packed):
            n = head[1 + b]
            out.append({int(row[i]): row[nmax + i] for i in range(n) if row[2 * nmax + i] > 0})
        return out

    # ---------------------------------------------------------------- interface
    def _postprocess_training(
        self, data: DecoderRawOutputType, batch: BatchType
    ) -> PostprocessingOutputType:
        output, side_outputs = data
        return {'instance_output': output, 'instance_side_outputs': side_outputs}

    def _denormalized_offset_scales(self, center_offset: torch.Tensor) -> Tuple[float, float]:
        if self._normalized_offset:
            h, w = center_offset.shape[-2:]
            return float(h), float(w)
        return 1.0, 1.0

    def _segment_with_foreground(self, center_heatmap, center_offset, foreground_mask):
        """grouping on the raw (normalised) offsets: the x H / x W of
        instance.py:361-365 happens inside the kernel (same fp32 rounding)."""
        sy, sx = self._denormalized_offset_scales(center_offset)
        return self._segment(center_heatmap, center_offset, foreground_mask, sy, sx)

    def _postprocess_inference(
        self, data: DecoderRawOutputType, batch: BatchType
    ) -> PostprocessingOutputType:
        output, side_outputs = data
        with_orientation = (len(output) == 3)
        if with_orientation:
            center_heatmap, center_offset, orientation = output
        else:
            center_heatmap, center_offset = output

        r = {
            'instance_output': output,
            'instance_side_outputs': side_outputs,
            'instance_centers': center_heatmap,
            'instance_offsets': center_offset,
        }
        if with_orientation:
            r['instance_orientation'] = orientation

        def _fullres(seg):
            crop, shape = get_valid_region_slices_and_fullres_shape(batch, 'instance')
            return self._crop_to_valid_region_and_resize_prediction(
                seg, valid_region_slices=crop, shape=shape, mode='nearest')

        # i-1: ground-truth foreground (dataset evaluation), instance.py:371-397
        if 'instance_foreground' in batch:
            fg = batch['instance_foreground'].to(center_heatmap.device)
            if fg.ndim == 4:
                fg = fg[:, 0]
            seg, meta = self._segment_with_foreground(center_heatmap, center_offset, fg)
            r['instance_segmentation_gt_foreground'] = seg
            r['instance_segmentation_gt_meta'] = meta
            r[get_fullres_key('instance_segmentation_gt_foreground')] = _fullres(seg)

        # i-2: everything foreground (debugging), instance.py:400-420
        if self.debug:
            fg_all = torch.ones_like(center_heatmap[:, 0], dtype=torch.bool)
            seg, _ = self._segment_with_foreground(center_heatmap, center_offset, fg_all)
            # the reference builds this mask with ones_like(center_heatmap): its debug result
            # keeps the channel axis, [B,1,H,W]
            seg = seg.unsqueeze(1)
            r['instance_segmentation_all_foreground'] = seg
            r[get_fullres_key('instance_segmentation_all_foreground')] = _fullres(seg)

        if not with_orientation:
            return r

        # o-1 / o-2 (instance.py:432-449), o-3 / o-4 debugging (:451-466)
        if all(k in batch for k in ('instance', 'orientation_foreground')):
            r['orientations_gt_instance_gt_orientation_foreground'] = \
                self._get_instance_orientation(orientation, batch['instance'].to(orientation.device),
                                               batch['orientation_foreground'].to(orientation.device))
        if all(k in batch for k in ('instance_foreground', 'orientation_foreground')):
            r['orientations_instance_segmentation_gt_orientation_foreground'] = \
                self._get_instance_orientation(orientation,
                                               r['instance_segmentation_gt_foreground'],
                                               batch['orientation_foreground'].to(orientation.device))
        if self.debug:
            r['orientations_gt_instance'] = self._get_instance_orientation(
                orientation, batch['instance'].to(orientation.device), None)
            r['orientations_instance_segmentation'] = self._get_instance_orientation(
                orientation, r['instance_segmentation_gt_foreground'], None)
        return r

