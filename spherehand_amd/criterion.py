"""Network wrapper and loss assembly -- the reference's
network/create_network_and_criterion.py surface:

    HeatmapEstimationNetwork   :27-144   hourglass + soft-argmax, real / synthetic / both
    MultiTaskLoss              :147-263  the loss terms and their weights
    average_joint_error        network/utils_metric.py:7-17
"""
import torch
import torch.nn as nn

from . import ops
from .hourglass import create_hourglass_network
from .multiview_utility import MultiviewConsistencyLoss, MutualProjectionLoss
from .render import BoneLengthLoss, CollisionLoss
from .util_modules import RecoverXYZCoordinateFromHeatmap, ResizeCropImage, TemporalSmoothnessLoss

SYNT_KEY_POINTS = [33, 32, 27, 26, 21, 20, 15, 14, 39, 40, 38, 0, 1, 2]   # network/constants.py:30
REAL_KEY_POINTS = [0, 3, 6, 9, 12, 15, 18, 21, 24, 25, 27, 30, 31, 32]    # network/constants.py:31


def average_joint_error(gt_joints, est_joints):
    """Mean over the 14 evaluation key-point pairs of ||gt - est||_2 (the NYU metric);
    returned as a 0-dim tensor on the inputs' device (no host sync)."""
    gt = gt_joints[:, :, REAL_KEY_POINTS, :].reshape(-1, len(REAL_KEY_POINTS), 3)
    est = est_joints[:, :, SYNT_KEY_POINTS, :].reshape(-1, len(SYNT_KEY_POINTS), 3)
    return (gt - est).norm(dim=-1).mean()


class HeatmapEstimationNetwork(nn.Module):
    """forward(real_dms[B,V,S,S]=None, synt_dms[N,S,S]=None) -> result dict with
    per-stack lists 'real_uv_hms' [B,V,J,h,w], 'real_d_hms', 'real_xyz' [B,V,J,3]
    and/or 'synt_uv_hms', 'synt_d_hms', 'synt_xyz'; with both inputs one
    concatenated pass (+ 'batch_*_fea' latents).  Training-time scale augmentation
    of the real crops as in the reference (:41-50, :93-102)."""

    def __init__(self, heatmap_size, depth_scale, num_joints, num_stacks, real_aug=True):
        super().__init__()
        self.num_joints = num_joints
        self.hg = create_hourglass_network(num_joints * 2, num_stacks)
        self.xyz_recover = RecoverXYZCoordinateFromHeatmap(heatmap_size, heatmap_size, depth_scale)
        self.resize_dm = ResizeCropImage() if real_aug else None

    def _augment(self, dms):
        n = dms.shape[0]
        if self.resize_dm is None or not self.training or torch.rand(1).item() < 0.5:
            return dms, None, None          # unit scales: nothing to undo on the way out
        scale = torch.rand(n, device=dms.device) * 0.2 + 0.75
        u_scale = scale + torch.rand_like(scale) * 0.1 - 0.05
        v_scale = scale + torch.rand_like(scale) * 0.1 - 0.05
        return self.resize_dm(dms, u_scale, v_scale), u_scale, v_scale

    def _split(self, outputs):
        return [o[:, :self.num_joints] for o in outputs], [o[:, self.num_joints:] for o in outputs]

    def _real_result(self, outputs, uv_hms, d_hms, u_scale, v_scale, B, V):
        xyz = [self.xyz_recover.from_output(o) for o in outputs]
        if u_scale is not None:
            inv = torch.stack([1.0 / u_scale, 1.0 / v_scale, torch.ones_like(u_scale)], dim=-1).unsqueeze(1)
            xyz = [p * inv for p in xyz]
        shape5 = lambda h: h.reshape(B, V, self.num_joints, h.shape[-2], h.shape[-1])   # noqa: E731
        return {'real_uv_hms': [shape5(h) for h in uv_hms], 'real_d_hms': [shape5(h) for h in d_hms],
                'real_xyz': [p.reshape(B, V, self.num_joints, 3) for p in xyz]}

    def forward(self, real_dms=None, synt_dms=None):
        result = {}
        n_synt = 0 if synt_dms is None else synt_dms.shape[0]
        inputs = [] if synt_dms is None else [synt_dms]
        if real_dms is not None:
            B, V = real_dms.shape[0], real_dms.shape[1]
            flat, u_scale, v_scale = self._augment(real_dms.reshape(B * V, real_dms.shape[2], real_dms.shape[3]))
            inputs.append(flat)
        outputs, latents = self.hg(torch.cat(inputs, dim=0) if len(inputs) > 1 else inputs[0])
        if synt_dms is not None:
            synt_out = [o[:n_synt] for o in outputs]
            uv, d = self._split(synt_out)
            result.update({'synt_uv_hms': uv, 'synt_d_hms': d,
                           'synt_xyz': [self.xyz_recover.from_output(o) for o in synt_out]})
        if real_dms is not None:
            real_out = [o[n_synt:] for o in outputs]
            uv, d = self._split(real_out)
            result.update(self._real_result(real_out, uv, d, u_scale, v_scale, B, V))
            if synt_dms is not None:
                if self.resize_dm is not None:
                    result['real_resized_dms'] = flat
                result['batch_synt_fea'] = [l[:n_synt] for l in latents]
                result['batch_real_fea'] = [l[n_synt:] for l in latents]
        return result


class MultiTaskLoss(nn.Module):
    """forward(result, synt_target=None, real_target=None) -> (loss_terms dict,
    projected depth maps per stack).  Same switches, terms and weights as the
    reference (:171-181).  `mesh` = hand model dict (or `constant` object with a
    .mesh attribute, as the reference passes); `prior_loss` = an optional module
    with .prior_loss(xyz/100) (the reference's frozen PoseVae; not part of this
    path -- pass an instance to enable the term)."""

    def __init__(self, synthesized_loss, mv_projection_loss, mv_consistency_loss, temporal_smooth_loss,
                 prior_loss, collision_loss, bone_length_loss, constant, image_size=64, heatmap_size=16):
        super().__init__()
        mesh = constant.mesh if hasattr(constant, 'mesh') else constant
        self.synthesized_loss = nn.MSELoss() if synthesized_loss else None
        self.mv_projection_loss = MutualProjectionLoss(image_size, mesh) if mv_projection_loss else None
        self.mv_consistency_loss = MultiviewConsistencyLoss() if mv_consistency_loss else None
        self.temporal_smooth_loss = TemporalSmoothnessLoss() if temporal_smooth_loss else None
        self.prior_loss = prior_loss if isinstance(prior_loss, nn.Module) else None
        self.collision_criterion = CollisionLoss() if collision_loss else None
        self.bone_length_criterion = BoneLengthLoss() if bone_length_loss else None
        if self.bone_length_criterion is not None:     # int32 / flat copies of its tables for the fused kernel
            bl = self.bone_length_criterion
            self.register_buffer('_bone_a', bl.joint_1.to(torch.int32), persistent=False)
            self.register_buffer('_bone_b', bl.joint_2.to(torch.int32), persistent=False)
            self.register_buffer('_bone_min', bl.min_length.reshape(-1).clone(), persistent=False)
            self.register_buffer('_bone_max', bl.max_length.reshape(-1).clone(), persistent=False)
        self.domain_loss = nn.MSELoss()
        self.heatmap_size = heatmap_size
        self._weight_cache = (None, None)
        self.weights = {'synt_hm': 1e3, 'synt_pt': 1e-1, 'mv_consistency': 1e-3, 'mv_projection': 1,
                        'temporal_smooth': 1.0, 'prior': 1e-2, 'hm_mean': 1e-2, 'domain': 0.0,
                        'collision': 1.0, 'bone_length': 1.0}

    def forward(self, result, synt_target=None, real_target=None):
        w, projected_dms = self.weights, []
        entries = []            # (term, weight, its value per hourglass stack); weighed and summed in one go below
        mse = self.synthesized_loss
        if mse is not None and synt_target is not None:
            entries.append(('synt_uv', w['synt_hm'], [mse(h, synt_target['uv_hms']) for h in result['synt_uv_hms']]))
            target_z = synt_target['xyz_pts'][:, :, 2]
            entries.append(('synt_d', w['synt_pt'], [mse(xyz[:, :, 2], target_z) for xyz in result['synt_xyz']]))
        is_mv = False if real_target is None else real_target.get('is_mv', True)
        if self.mv_projection_loss is not None and real_target is not None:
            values = []
            for xyz in result['real_xyz']:
                loss, dm = self.mv_projection_loss(real_target['camera_poses'], real_target['inv_camera_poses'], xyz,
                                                   real_target['real_dms'], is_mv)
                values.append(loss)
                projected_dms.append(dm)
            entries.append(('mv_projection', w['mv_projection'], values))
            self.mv_projection_loss.invalidate()     # (the stacks shared one compaction; nothing stays pinned between steps)
        if self.mv_consistency_loss is not None and real_target is not None:
            entries.append(('mv_consistency', w['mv_consistency'] if is_mv else 0,
                            [self.mv_consistency_loss(real_target['camera_poses'], xyz, None)
                             for xyz in result['real_xyz']]))
        if real_target is not None:
            # like the reference this term needs the MSE criterion of the synthetic switch (:235)
            # MSELoss(h, zeros_like(h)) = mean(h^2): the same value without the zero image and the subtraction
            entries.append(('uv_hm_mean', w['hm_mean'], [h.square().mean() for h in result['real_uv_hms']]))
        real_xyz = result.get('real_xyz', [])
        if self.prior_loss is not None:
            entries.append(('pose_prior', w['prior'], [self.prior_loss.prior_loss(xyz / 100.0) for xyz in real_xyz]))
        if self.temporal_smooth_loss is not None:
            entries.append(('temporal_smooth', w['temporal_smooth'],
                            [self.temporal_smooth_loss(xyz) for xyz in real_xyz]))
        cc, bc = self.collision_criterion, self.bone_length_criterion
        if cc is not None and bc is not None and real_xyz and real_xyz[0].is_cuda and real_xyz[0].dtype == torch.float32:
            # both hinge losses and their gradients in one launch (same indexing as the modules: the first 41
            # points of joints.view(B, -1, 3))
            pairs = [ops.PairLosses.apply(xyz.reshape(xyz.shape[0], -1, 3), 41, 11, 6, float(cc.min_sq_dist),
                                          self._bone_a, self._bone_b, self._bone_min, self._bone_max) for xyz in real_xyz]
            entries.append(('collision', w['collision'], [p[0] for p in pairs]))
            entries.append(('bone_length', w['bone_length'], [p[1] for p in pairs]))
        else:
            if cc is not None:
                entries.append(('collision', w['collision'], [cc(xyz) for xyz in real_xyz]))
            if bc is not None:
                entries.append(('bone_length', w['bone_length'], [bc(xyz) for xyz in real_xyz]))
        if 'batch_synt_fea' in result and 'batch_real_fea' in result:
            if w['domain'] == 0.0:
                # the reference's weight is 0 (create_network...:180): the term is reported as 0 and has no gradient;
                # evaluating the two feature means and their MSE for it cost ~16 launches per step
                entries.append(('domain_loss', 0.0, None))
            else:
                entries.append(('domain_loss', w['domain'],
                                [self.domain_loss(s.mean(dim=(0, 2, 3)), r.mean(dim=(0, 2, 3)))
                                 for s, r in zip(result['batch_synt_fea'], result['batch_real_fea'])]))
        return self._weigh(entries), projected_dms

    def _weigh(self, entries):
        """term = sum over the stacks of weight * value (the reference's sums at :196-262), evaluated for all terms
        at once: one stack, one multiply by the cached weight vector, one row sum -- instead of a multiply and an add
        per term and stack (and as many again in the backward).  Returns a LossTerms dict whose `.stacked` is the [K]
        tensor the values are views of."""
        live = [(name, wt, vals) for name, wt, vals in entries if vals]
        depth = {len(vals) for _, _, vals in live}
        if not live or len(depth) != 1:         # no tensors, or ragged stacks: term by term
            return LossTerms((name, sum(wt * v for v in vals) if vals else 0) for name, wt, vals in entries)
        S, ref = depth.pop(), live[0][2][0]
        zero = None
        rows, weights = [], []
        for name, wt, vals in entries:
            if not vals:                        # reported as 0: a row of zeros with weight 0
                zero = ref.new_zeros(()) if zero is None else zero
                vals = [zero] * S
                wt = 0.0
            rows.extend(v.reshape(()) for v in vals)
            weights.append(float(wt))
        key = (tuple(weights), ref.device, ref.dtype)
        if self._weight_cache[0] != key:
            self._weight_cache = (key, torch.tensor(weights, device=ref.device, dtype=ref.dtype).unsqueeze(1))
        stacked = (torch.stack(rows).view(len(entries), S) * self._weight_cache[1]).sum(dim=1)
        terms = LossTerms(zip((name for name, _, _ in entries), stacked.unbind(0)))
        terms.stacked = stacked
        return terms


class LossTerms(dict):
    """The loss terms by name; `.stacked` (when set) is the [K] tensor holding them in order, so the total and the
    running averages need no second stack."""
    stacked = None


def stack_terms(loss_terms):
    """The terms as one [K] tensor (python numbers become constants): one launch feeds both the total loss and the
    running averages."""
    stacked = getattr(loss_terms, 'stacked', None)
    if stacked is not None and stacked.numel() == len(loss_terms):
        return stacked
    vals = list(loss_terms.values())
    ref = next((v for v in vals if torch.is_tensor(v)), None)
    if ref is None:
        return torch.tensor([float(v) for v in vals])
    return torch.stack([v.reshape(()).to(ref.dtype) if torch.is_tensor(v) else ref.new_tensor(float(v)) for v in vals])


def combine_loss(loss_terms, stacked=None):
    """Sum of the terms (network/engine.py: sum_loss_terms); `stacked` = stack_terms(loss_terms) if already made."""
    return (stack_terms(loss_terms) if stacked is None else stacked).sum()
