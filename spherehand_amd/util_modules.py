"""Network-side helpers around the render path -- same classes and forward
signatures as the reference's network/util_modules.py.

    HandSynthesizer                     network/util_modules.py:86-122
    RecoverXYZCoordinateFromHeatmap     :164-201   (soft-argmax heat-map -> xyz)
    HeatmapVariance                     :204-240
    DepthNoise / DepthResample          :46-84 / :10-43
    ResizeCropImage                     :383-424
    TemporalSmoothnessLoss              :367-381
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .kinematicsTransformation import HandTransformationMat
from .pointTransformation import RandScale
from .render import DepthRender, Hand3DHeatmapRender


def _spatial_softmax(hms, sigma):
    n, j, h, w = hms.shape
    return F.softmax((hms * sigma).reshape(n * j, h * w), dim=1).reshape(n, j, h, w)


def _spatial_normalize(hms):
    """relu(h) / (sum relu(h) + 1e-5) per map (network/util_modules.py:144-161)."""
    h = torch.relu(hms)
    return h / (h.sum(dim=(-2, -1), keepdim=True) + 1e-5)


class RecoverXYZCoordinateFromHeatmap(nn.Module):
    """Soft-argmax: uv = sum softmax(20*uv_hm) * grid; depth = sum d_hm * relu-normalised
    uv_hm; then through the inverse orthographic camera of the heat-map
    (x = (u - W/2) / (W/300), z = d / depth_scale)."""

    def __init__(self, width, height, depth_scale):
        super().__init__()
        self.depth_scale = 1.0 / depth_scale
        self.fx, self.fy = width / 300.0, height / 300.0
        self.cx, self.cy = width / 2, height / 2
        # full [1,1,H,W] grids: they are part of network_state_dict (HeatmapEstimationNetwork.xyz_recover),
        # so the shapes are the reference's (network/util_modules.py:174-181) and checkpoints interoperate
        u = torch.arange(width, dtype=torch.float32).view(1, 1, 1, width).expand(1, 1, height, width)
        v = torch.arange(height, dtype=torch.float32).view(1, 1, height, 1).expand(1, 1, height, width)
        self.register_buffer('u_grid', u.contiguous())
        self.register_buffer('v_grid', v.contiguous())

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # checkpoints written by this repository before round 2 hold the grids as [1,1,1,W] / [1,1,H,1]: same values
        for name in ('u_grid', 'v_grid'):
            t = state_dict.get(prefix + name)
            if t is not None and t.shape != getattr(self, name).shape and t.numel() in (self.u_grid.shape[-1], self.u_grid.shape[-2]):
                state_dict[prefix + name] = t.expand_as(getattr(self, name)).contiguous()
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def from_output(self, hm):
        """xyz from the network's raw output hm [N,2J,h,w] (uv maps | depth maps): one kernel per direction on
        the GPU, the torch formulation (forward()) otherwise."""
        J = hm.shape[1] // 2
        if ops.soft_argmax_supported(hm, J):
            return ops.SoftArgmaxXYZ.apply(hm, J, self.cx, self.cy, self.fx, self.fy, self.depth_scale)
        return self.forward(hm[:, :J], hm[:, J:])

    def forward(self, uv_hms, d_hms, is_shuffing=False):
        p = _spatial_softmax(uv_hms, 20.0)
        u = (p * self.u_grid).sum(dim=(-2, -1))
        v = (p * self.v_grid).sum(dim=(-2, -1))
        d = (d_hms * _spatial_normalize(uv_hms)).sum(dim=(-2, -1))
        return torch.stack([(u - self.cx) / self.fx, (v - self.cy) / self.fy, d * self.depth_scale], dim=-1)


class HeatmapVariance(nn.Module):
    """Spread of a heat-map about its soft-argmax (sigma 25), in normalised image units."""

    def __init__(self, width, height):
        super().__init__()
        u = (torch.arange(width, dtype=torch.float64) - width / 2) / width
        v = (torch.arange(height, dtype=torch.float64) - height / 2) / height
        self.register_buffer('u_grid', u.float().view(1, 1, 1, width))
        self.register_buffer('v_grid', v.float().view(1, 1, height, 1))

    def forward(self, hms):
        p = _spatial_softmax(hms, 25.0)
        w = _spatial_normalize(hms)
        out = 0
        for grid in (self.u_grid, self.v_grid):
            mean = (p * grid).sum(dim=(-2, -1), keepdim=True)
            out = out + (w * (grid - mean) ** 2).sum(dim=(-2, -1))
        return out


class DepthResample(nn.Module):
    """Random drop-out to 1.0 (probability 1 - sample_ratio) followed by a fixed 3x3 /
    5x5 Gaussian blur.  RNG: torch.rand_like on the input's device."""
    _K3 = [1, 2, 1, 2, 6, 2, 1, 2, 1]
    _K5 = [1, 4, 7, 4, 1, 4, 16, 26, 16, 4, 7, 26, 41, 26, 7, 4, 16, 26, 16, 4, 1, 4, 7, 4, 1]

    def __init__(self, sample_ratio, kernel_size=3):
        super().__init__()
        self.sample_ratio = sample_ratio
        k = torch.tensor(self._K5 if kernel_size == 5 else self._K3).float()
        self.gaussian_filter = nn.Conv2d(1, 1, kernel_size, padding=kernel_size // 2, bias=False)
        self.gaussian_filter.weight.data = (k / k.sum()).view(1, 1, kernel_size, kernel_size)
        self.gaussian_filter.weight.requires_grad = False

    def forward(self, dm):
        if dm.ndimension() == 3:
            dm = dm.unsqueeze(1)
        if dm.is_cuda and dm.dtype == torch.float32 and dm.shape[1] == 1 and not (torch.is_grad_enabled() and dm.requires_grad):
            # one torch.rand + one launch (drop-out and the FIXED 3x3 / 5x5 Gaussian of the reference fused: the kernel
            # carries those tables -- a changed self.gaussian_filter.weight takes the torch path below)
            w = self.gaussian_filter.weight
            # a trainable filter always takes the torch path; a frozen one is compared with the module's table once per
            # weight version (the table is cached on the weight's device: no host-to-device copy and no
            # synchronising torch.equal per forward, both of which would also break a hipGraph capture).  An
            # assignment through `weight.data = ...` does not bump the version counter: callers that swap the kernel
            # that way must reset `_fixed_kernel_checked` (the reference never does, network/util_modules.py:10-43).
            if w.requires_grad:
                self._fixed = False
            elif getattr(self, '_fixed_kernel_checked', None) != (w._version, w.device):
                k = getattr(self, '_fixed_table', None)
                if k is None or k.device != w.device or k.numel() != w.numel():
                    k = torch.tensor(self._K5 if w.shape[-1] == 5 else self._K3, dtype=w.dtype, device=w.device)
                    k = self._fixed_table = k / k.sum()
                self._fixed = bool(torch.equal(w.reshape(-1), k))
                self._fixed_kernel_checked = (w._version, w.device)
            if not self._fixed:
                dm = torch.where(torch.rand_like(dm) > self.sample_ratio, torch.ones_like(dm), dm)
                return self.gaussian_filter(dm)
            return ops.depth_resample(dm.reshape(dm.shape[0], dm.shape[2], dm.shape[3]).contiguous(), self.sample_ratio,
                                      self.gaussian_filter.kernel_size[0])
        dm = torch.where(torch.rand_like(dm) > self.sample_ratio, torch.ones_like(dm), dm)
        return self.gaussian_filter(dm)


class DepthNoise(nn.Module):
    """Per-pixel random source shift (sigma 0.5 px, rounded) and Gaussian z noise
    (sigma 0.05) on foreground pixels (< 1.0 in scaled depth)."""

    def __init__(self, width, height):
        super().__init__()
        self.sigma_x = self.sigma_y = 0.5
        self.sigma_z = 0.05
        self.register_buffer('u_grid', torch.arange(width).view(1, 1, width))
        self.register_buffer('v_grid', torch.arange(height).view(1, height, 1))

    def forward(self, dm):
        n, h, w = dm.shape
        if dm.is_cuda and dm.dtype == torch.float32 and not torch.is_grad_enabled():
            return ops.depth_noise(dm.contiguous(), self.sigma_x, self.sigma_z)       # one randn + one launch
        sx = torch.clamp((torch.randn_like(dm) * self.sigma_x + 0.5).long() + self.u_grid, 0, w - 1)
        sy = torch.clamp((torch.randn_like(dm) * self.sigma_y + 0.5).long() + self.v_grid, 0, h - 1)
        noisy = torch.gather(dm.reshape(n, h * w), 1, (sy * w + sx).reshape(n, h * w)).view(n, h, w)
        return torch.where(noisy < 1.0, noisy + torch.randn_like(noisy) * self.sigma_z, noisy)


class ResizeCropImage(nn.Module):
    """Per-image anisotropic down-scale (nearest neighbour) pasted centred on a canvas of
    ones; like the reference, only scales <= 1 in v are pasted (its paste sits under the
    v-branch's else, network/util_modules.py:411-423).  One batched gather on the device,
    no per-image loop and no host synchronisation (the reference resizes image by image,
    :392-423); `forward_loop` keeps that formulation for the tests."""

    def forward(self, depth_maps, u_scales, v_scales):
        n, h, w = depth_maps.shape
        dev = depth_maps.device
        us, vs = u_scales.to(dev).double(), v_scales.to(dev).double()        # python-float arithmetic of the reference
        new_h, new_w = (h * vs + 0.5).long(), (w * us + 0.5).long()          # int(x + 0.5)
        big_u = us > 1
        ou0 = torch.where(big_u, (new_w - w) // 2, torch.zeros_like(new_w))  # first resized column that is pasted
        ncol = torch.where(big_u, torch.full_like(new_w, w), (w * us).long())
        u0 = torch.where(big_u, torch.zeros_like(new_w), (w - new_w) // 2)   # first canvas column
        nrow = (h * vs).long()
        v0 = (h - new_h) // 2
        rows = torch.arange(h, device=dev).view(1, h)
        cols = torch.arange(w, device=dev).view(1, w)
        r = rows - v0.view(n, 1)                                              # row in the resized image
        c = cols - u0.view(n, 1) + ou0.view(n, 1)
        row_ok = (r >= 0) & (r < nrow.view(n, 1)) & (vs <= 1).view(n, 1)
        col_ok = (cols >= u0.view(n, 1)) & (cols < (u0 + ncol).view(n, 1))
        # ATen nearest: src = min(floor(dst * (float)in/out), in - 1)
        sr = torch.clamp((r.float() * (h / new_h.clamp(min=1).float()).view(n, 1)).floor().long(), 0, h - 1)
        sc = torch.clamp((c.float() * (w / new_w.clamp(min=1).float()).view(n, 1)).floor().long(), 0, w - 1)
        picked = depth_maps.gather(1, sr.view(n, h, 1).expand(n, h, w)).gather(2, sc.view(n, 1, w).expand(n, h, w))
        return torch.where(row_ok.view(n, h, 1) & col_ok.view(n, 1, w), picked, torch.ones_like(depth_maps))

    def forward_loop(self, depth_maps, u_scales, v_scales):
        h, w = depth_maps.shape[-2], depth_maps.shape[-1]
        out = torch.ones_like(depth_maps)
        for idx, (us, vs) in enumerate(zip(u_scales.tolist(), v_scales.tolist())):
            if vs > 1:
                continue
            new_h, new_w = int(h * vs + 0.5), int(w * us + 0.5)
            resized = F.interpolate(depth_maps[idx].view(1, 1, h, w), (new_h, new_w))[0, 0]
            if us > 1:
                ou0 = (new_w - w) // 2
                u0, u1, ou1 = 0, w, ou0 + w
            else:
                ou0, ou1 = 0, int(w * us)
                u0 = (w - new_w) // 2
                u1 = u0 + ou1
            ov1 = int(h * vs)
            v0 = (h - new_h) // 2
            out[idx, v0:v0 + ov1, u0:u1] = resized[0:ov1, ou0:ou1]
        return out


class TemporalSmoothnessLoss(nn.Module):
    """Clamped (+-2500) L2 between consecutive samples' joints; remembers the last
    sample of the previous batch (stateful, like the reference)."""

    def __init__(self):
        super().__init__()
        self.previous_skel = None
        self.thresh = 2500.0

    def forward(self, joints):
        assert joints.ndimension() == 4
        if self.previous_skel is None:
            prev, curr = joints[:-1].detach(), joints[1:]
        else:
            prev = torch.cat([self.previous_skel.unsqueeze(0), joints[:-1].detach()], dim=0)
            curr = joints
        self.previous_skel = joints[-1].detach().clone()
        return (torch.clamp(prev - curr, -self.thresh, self.thresh) ** 2).mean()


class HandSynthesizer(nn.Module):
    """pose [B,26] -> (noisy scaled depth crop [B,S,S], uv heat-maps, depth heat-maps,
    key-point xyz), all detached: the synthetic training branch (square images).

    On the GPU the whole of network/util_modules.py:104-122 is THREE launches with nothing in between (`fused`, the
    default; capturable in a hipGraph): forward kinematics + RandScale + the samples' random draws
    (ops.synth_pose), skinning + camera + triangle raster + clamp + resize + `* depth_scale` + DepthNoise
    (ops.mesh_render_post: the noise runs in the rasterizer's epilogue), key-point skinning + heat-map camera + paint
    (ops.heatmap_render).  The random numbers come from a counter-based generator inside the kernels (synth_rng.py
    restates it): `rng_state` = (seed, call counter, ticket) lives on the device, seed = torch.initial_seed() + seed_offset at
    the first call -- and again whenever torch.manual_seed() has CHANGED it since; reseed() restarts the stream by hand.  Parity with
    the reference's torch.rand / randn draws is in distribution; with the same draws (`last_draws`) and the noise off
    the outputs equal the module-by-module chain bit for bit.  fused = False: that chain -- RandScale's three CPU
    draws, torch.rand for the focal jitter, DepthRender, DepthNoise on one torch.randn, Hand3DHeatmapRender."""

    def __init__(self, mesh, image_size, heatmap_size, uv_hm_scale, depth_scale, add_noise=True, out_heatmap=True):
        super().__init__()
        self.uv_hm_scale = uv_hm_scale
        self.depth_scale = depth_scale
        self.hand_skeleton_transform = HandTransformationMat(
            [bone['offset_matrix'].astype(np.float32) for bone in mesh['bones']])
        self.hm_render = Hand3DHeatmapRender(mesh['bones'], heatmap_size)
        self.dm_render = DepthRender(mesh, image_size)
        self.rand_scale = RandScale(0.1)
        self.depth_noiser = DepthNoise(image_size, image_size)
        self.add_noise = add_noise
        self.out_heatmap = out_heatmap
        self.fused = True
        self.one_launch = True       # the whole forward as ONE kernel where its sizes allow (ops.hand_synth)
        self.rng_state = None        # int64 [4] on the device: (seed, call counter, the launch's ticket, unused)
        self.seed_offset = 0         # added to torch.initial_seed(): Engine sets the rank, so that ranks seeded alike draw differently
        self._rng_seed = None
        self._seed_explicit = None   # set by reseed(seed): that seed instead of torch's
        self._rng_counter = 0        # the call counter the next (re)seeding starts from
        self.last_draws = None       # [6,B] of the last fused call: s_x, s_y, s_z, focal jitter, the two noise keys (bits)
        self._kp_bone_i32 = None

    def _default_seed(self):
        return torch.initial_seed() + int(self.seed_offset)

    def reseed(self, seed=None, device=None, counter=0):
        """Restart the kernels' random stream at call `counter`.  seed = None: follow torch -- torch.initial_seed() +
        seed_offset, re-read whenever torch.manual_seed() changes it; an explicit seed stays until the next reseed()."""
        self._seed_explicit = None if seed is None else int(seed)
        self._rng_counter = int(counter)
        dev = device if device is not None else (self.rng_state.device if self.rng_state is not None else None)
        self.rng_state = None
        if dev is not None:
            self._materialise(dev)

    def _materialise(self, dev):
        seed = self._default_seed() if self._seed_explicit is None else self._seed_explicit
        self._rng_seed = seed
        s64 = seed & (2 ** 64 - 1)
        self.rng_state = torch.tensor([s64 - 2 ** 64 if s64 >= 2 ** 63 else s64, self._rng_counter, 0, 0], dtype=torch.int64, device=dev)
        self._rng_counter = 0

    def _state(self, dev):
        stale = self._seed_explicit is None and self._rng_seed != self._default_seed()      # torch was re-seeded
        if self.rng_state is None or self.rng_state.device != dev or stale:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("HandSynthesizer: call the module (or reseed(device=...)) once before capturing it: "
                                   "seeding the generator state is a host-to-device copy")
            self._materialise(dev)
        return self.rng_state

    def _fused_ok(self, parameters):
        ras, nz = self.dm_render.rasterizer, self.depth_noiser
        return (self.fused and parameters.is_cuda and ras.fused and ras.width == ras.height and 2 * ras.width <= 641
                and nz.sigma_x == nz.sigma_y and 0.0 < nz.sigma_x <= 0.6)

    @torch.no_grad()
    def forward(self, parameters):
        if self._fused_ok(parameters):
            return self._forward_fused(parameters)
        transform_mats = self.rand_scale(self.hand_skeleton_transform(parameters))
        # focal jitter U(0.9, 1.1) (util_modules.py:110), drawn on the tensors' device (no host round trip)
        rand_f_ratio = torch.rand(transform_mats.shape[0], device=transform_mats.device) * 0.2 + 0.9
        depth = self.dm_render(transform_mats, rand_f_ratio) * self.depth_scale
        if self.add_noise:
            depth = self.depth_noiser(depth)
        if not self.out_heatmap:
            return depth
        uv_hms, depth_hms, xyz_pts = self.hm_render(transform_mats, rand_f_ratio, self.uv_hm_scale, self.depth_scale)
        return depth, uv_hms, depth_hms, xyz_pts

    def _forward_fused(self, parameters):
        dev = parameters.device
        state = self._state(dev)
        fk, dr, hm, nz = self.hand_skeleton_transform, self.dm_render, self.hm_render, self.depth_noiser
        ras, lbs = dr.rasterizer, dr.lbs
        heat = None
        if self.out_heatmap:
            kl = hm.lbs
            if self._kp_bone_i32 is None or self._kp_bone_i32.device != dev:
                self._kp_bone_i32 = kl.skin_bone.to(device=dev, dtype=torch.int32).contiguous()
                k = hm.inv_camera.inv_k_mat[0].cpu().tolist()
                self._inv_k = (float(k[0][0]), float(k[0][3]), float(k[1][1]), float(k[1][3]))
            cam = hm.camera
            hcam = (cam.cx, cam.cy, cam.fx, cam.fy)
            heat = (kl.skin_vertex_start, self._kp_bone_i32, kl.skin_wv, hm.width, hcam, hm.hm_renderer.sigma, self._inv_k,
                    self.uv_hm_scale, self.depth_scale)
        # (the one-launch kernel flips x once for both tables: the mesh's and the key-points' skinning are for one hand)
        if self.one_launch and (heat is None or hm.lbs.right_hand == lbs.right_hand):
            out = ops.hand_synth(parameters, fk.offset, fk.offset_inv, state, self.rand_scale.rand_scale, lbs, ras.faces_i32,
                                 dr.camera, ras.height, self.depth_scale, self.add_noise, nz.sigma_x, nz.sigma_z, heat)
            if out is not None:
                self.last_draws = out[0]
                return out[1:] if self.out_heatmap else out[1]
        # three launches: sizes the one-launch kernel does not take (no lattice of sampled pixels: S = 256 ...)
        T, draws = ops.synth_pose(parameters, fk.offset, fk.offset_inv, state, self.rand_scale.rand_scale)
        self.last_draws = draws
        rand_f = draws[3]
        depth = ops.mesh_render_post(T, lbs.skin_vertex_start, lbs.skin_bone, lbs.skin_wv, lbs.right_hand, dr.camera, rand_f,
                                     ras.faces_i32, ras.height, self.depth_scale, draws[4:6] if self.add_noise else None,
                                     nz.sigma_x, nz.sigma_z, state)
        if not self.out_heatmap:
            return depth
        uv_hms, depth_hms, xyz_pts = ops.heatmap_render(T, heat[0], heat[1], heat[2], hm.lbs.right_hand, heat[4], rand_f, heat[3],
                                                        heat[5], heat[6], heat[7], heat[8])
        return depth, uv_hms, depth_hms, xyz_pts

