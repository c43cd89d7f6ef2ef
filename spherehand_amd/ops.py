"""Tensor-level entry points over the C ABI (device memory + stream plumbing only).

Every function takes CUDA(HIP) fp32 contiguous tensors, allocates its outputs
with torch, and enqueues the hand-written HIP kernels on torch's CURRENT stream.
Preconditions mirror the reference extension's CHECK_INPUT
(mesh/cuda_kernel/depth_rasterization_cuda.cpp:11-19): a violation raises
RuntimeError.
"""
import torch

from . import _lib


def _ptr(t):
    return None if t is None else t.data_ptr()


# (device index) -> hipStream_t of torch's current stream, and the current device index.  The two private accessors
# are 13 us per call cheaper than the public spelling (see _stream); a torch build without them (CPU-only wheels, a
# rename) falls back to the public one -- importing this module never fails for that.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda d: torch.cuda.current_stream(d).cuda_stream)
_cur_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def _stream():
    """torch's current HIP stream on the current device, as the integer the C ABI takes.  (The public spelling
    torch.cuda.current_stream().cuda_stream builds a Stream object through three Python layers: 13 us per call, six
    calls per pose -> depth -> pose iteration -- a third of its eager time, tools/prof_eager.py.)"""
    return _raw_stream(_cur_device())


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on(device):
    """Device guard for the launches below: a no-op when `device` is already current (one process
    per GPU: always), torch.cuda.device(...) otherwise."""
    if device.index is None or device.index == _cur_device():
        return _NO_SWITCH
    return torch.cuda.device(device)


def _check_input(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s" % (name, dtype))


RASTER_OWNER_TOUCHED_ROWS = 1   # shr_sphere_raster_fwd_ex flag (include/spherehand_hip.h)


def sphere_raster_fwd(spheres, H, W, want_argmin=False, flags=0):
    """spheres [N,J,4] (x,y,z,r) -> depth [N,H,W] (and uint8 argmin [N,H,W]).  flags = RASTER_OWNER_TOUCHED_ROWS:
    the owner bytes of rows no sphere touches stay unwritten (all sphere_raster_bwd needs)."""
    _check_input(spheres, "spheres")
    if spheres.dim() != 3 or spheres.shape[2] != 4:
        raise RuntimeError("spheres must be [N,J,4]")
    N, J, _ = spheres.shape
    with _on(spheres.device):
        depth = torch.empty((N, H, W), dtype=torch.float32, device=spheres.device)
        arg = torch.empty((N, H, W), dtype=torch.uint8, device=spheres.device) if want_argmin else None
        _lib.check(_lib.lib().shr_sphere_raster_fwd_ex(_ptr(spheres), N, J, H, W, _ptr(depth), _ptr(arg), int(flags),
                                                       _stream()), "shr_sphere_raster_fwd")
    return (depth, arg) if want_argmin else depth


def sphere_raster_bwd(spheres, grad_depth, argmin=None):
    """grad_depth [N,H,W] (+ the forward's uint8 owner map, or None to recompute
    the owners) -> grad_spheres [N,J,4]."""
    _check_input(spheres, "spheres")
    _check_input(grad_depth, "grad_depth")
    N, J, _ = spheres.shape
    if grad_depth.dim() != 3 or grad_depth.shape[0] != N:
        raise RuntimeError("grad_depth must be [N,H,W]")
    H, W = grad_depth.shape[1], grad_depth.shape[2]
    if argmin is not None:
        _check_input(argmin, "argmin", torch.uint8)
        if argmin.shape != grad_depth.shape:
            raise RuntimeError("argmin must be [N,H,W]")
    with _on(spheres.device):
        out = torch.empty((N, J, 4), dtype=torch.float32, device=spheres.device)
        _lib.check(_lib.lib().shr_sphere_raster_bwd(_ptr(spheres), _ptr(grad_depth), _ptr(argmin), N, J, H, W,
                                                    _ptr(out), _stream()), "shr_sphere_raster_bwd")
    return out


TUNE_FWD_LDS_BYTES, TUNE_FWD_OWNER_LDS_BYTES, TUNE_BWD_LDS_BYTES, TUNE_FORCE_GENERAL, TUNE_FWD_WAVES = 1, 2, 3, 4, 5
TUNE_FWD_SHARES, TUNE_BWD_SHARES, TUNE_D2M_WAVES, TUNE_D2M_BAND_UNITS, TUNE_PERSISTENT, TUNE_FWD_ZBUF_BYTES = 6, 7, 8, 9, 10, 11
TUNE_BWD_WAVES, TUNE_MSE_BOX, TUNE_FWD_RUN_TABLE, TUNE_D2M_TILED, TUNE_TRI_BAND, TUNE_MESH_BAND = 12, 13, 14, 15, 16, 17


def set_tuning(key, value):
    """Launch-shape / test hook (shr_set_tuning); never changes results."""
    _lib.check(_lib.lib().shr_set_tuning(int(key), int(value)), "shr_set_tuning")


class SphereDepthRaster(torch.autograd.Function):
    """depth[N,H,W] = min over the J spheres of a crop (SURVEY 8b "new
    differentiable op").  Differentiable w.r.t. spheres[N,J,4] = (x,y,z,r).
    The forward saves its uint8 owner map (1 byte/pixel) for the backward -- written on the rows the backward
    will look at only (RASTER_OWNER_TOUCHED_ROWS: the map never leaves this Function)."""

    @staticmethod
    def forward(ctx, spheres, H, W):
        spheres = spheres.contiguous()
        if ctx.needs_input_grad[0]:
            depth, owner = sphere_raster_fwd(spheres, H, W, want_argmin=True, flags=RASTER_OWNER_TOUCHED_ROWS)
            ctx.save_for_backward(spheres, owner)
        else:
            depth = sphere_raster_fwd(spheres, H, W)
        return depth

    @staticmethod
    def backward(ctx, grad_depth):
        spheres, owner = ctx.saved_tensors
        return sphere_raster_bwd(spheres, grad_depth.contiguous(), owner), None, None


def _check_index(index, n, m, name):
    if index.dtype != torch.int32 or index.dim() != 1 or index.numel() != n or not index.is_cuda or not index.is_contiguous():
        raise RuntimeError("%s must be a contiguous int32 CUDA tensor with one entry per crop" % name)
    # PRECONDITION (not checked here -- reading the values back would synchronise the stream): every entry is an
    # image number in [0, m).  The kernels index the image stack with it unchecked; MutualProjectionLoss builds
    # and caches its index tensors once per (B, V, device), in range by construction.


def sphere_raster_mse_supported(spheres, target, H, W):
    """True when shr_sphere_raster_mse takes these buffers (16-byte rows, image fits the kernel)."""
    return (W % 4 == 0 and spheres.data_ptr() % 16 == 0 and target.data_ptr() % 16 == 0
            and _lib.lib().shr_sphere_raster_mse_regions(int(H), int(W)) > 0)


def sphere_raster_mse(spheres, target, target_index=None, want_depth=True):
    """spheres [N,J,4], target [M,H,W] (crop n compares with image target_index[n], or n)
    -> (depth [N,H,W] or None, sse [N], grad_spheres [N,J,4]) with sse[n] = sum over the
    crop of (depth - target)^2 and grad_spheres = d sse[n] / d (x,y,z,r): the fused
    render-and-compare kernel, no owner map / gradient image in HBM."""
    _check_input(spheres, "spheres")
    _check_input(target, "target")
    if spheres.dim() != 3 or spheres.shape[2] != 4 or target.dim() != 3:
        raise RuntimeError("spheres must be [N,J,4] and target [M,H,W]")
    N, J, _ = spheres.shape
    H, W = target.shape[1:]
    if target_index is None and target.shape[0] != N:
        raise RuntimeError("target must hold one image per crop unless target_index is given")
    if target_index is not None:
        _check_index(target_index, N, target.shape[0], "target_index")
    R = _lib.lib().shr_sphere_raster_mse_regions(int(H), int(W))
    if R <= 0:
        raise RuntimeError("image rows too wide for the fused kernel (compose sphere_raster_fwd / _bwd)")
    with _on(spheres.device):
        depth = torch.empty((N, H, W), dtype=torch.float32, device=spheres.device) if want_depth else None
        sse = torch.empty((N, R), dtype=torch.float32, device=spheres.device)
        grad = torch.empty((N, R, J, 4), dtype=torch.float32, device=spheres.device)
        _lib.check(_lib.lib().shr_sphere_raster_mse(_ptr(spheres), N, J, H, W, _ptr(target), _ptr(target_index),
                                                    _ptr(depth), _ptr(sse), _ptr(grad), _stream()),
                   "shr_sphere_raster_mse")
        if R > 1:
            sse, grad = sse.sum(1), grad.sum(1)
        else:
            sse, grad = sse.view(N), grad.view(N, J, 4)
    return depth, sse, grad


class SphereRasterSSE(torch.autograd.Function):
    """(spheres [N,J,4], target [M,H,W], target_index [N] int32 or None) -> (sse [N], depth
    [N,H,W]): per-crop sum of squared differences between the rendered spheres and an
    observed depth image, with the rendered depth as a second, non-differentiable output.
    The backward only scales the gradient the fused kernel already produced."""

    @staticmethod
    def forward(ctx, spheres, target, target_index=None):
        spheres = spheres.contiguous()
        depth, sse, grad = sphere_raster_mse(spheres, target.contiguous(), target_index)
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(depth)
        ctx.set_materialize_grads(False)    # no zero image for the depth output's "gradient"
        return sse, depth

    @staticmethod
    def backward(ctx, grad_sse, _grad_depth):
        (grad,) = ctx.saved_tensors
        if grad_sse is None:
            return None, None, None
        return grad * grad_sse.view(-1, 1, 1), None, None


D2M_TWO_STEP = True   # data->model: compact the images once + search the point lists (False: the streaming kernel)


def data_to_model(depth, centres, radii, want_grad=False, depth_index=None):
    """depth [N,H,W], centres [N,J,3], radii [J] -> loss_sum [N] (and the unit
    gradient d loss_sum[n]/d centres [N,J,3]).  With depth_index [N] int32, depth is
    [M,H,W] and crop n reads image depth_index[n].
    Two paths by stack size (d2m_two_step_pays): the two-step path returns ONE integer sum per crop; the streaming
    kernel returns R = shr_data_to_model_parts partial sums per crop (R = 2 from 192x192 pixels on), each an exact
    integer sum, ADDED here in fp32.  The two agree bit for bit where R = 1 and to one fp32 rounding of the crop's
    total where R = 2 -- a crop's value may move by that rounding with the batch size that selects the path."""
    _check_input(depth, "depth")
    _check_input(centres, "centres")
    _check_input(radii, "radii")
    if depth.dim() != 3 or centres.dim() != 3 or centres.shape[2] != 3:
        raise RuntimeError("depth must be [N,H,W] and centres [N,J,3]")
    if depth_index is None and centres.shape[0] != depth.shape[0]:
        raise RuntimeError("depth must be [N,H,W] and centres [N,J,3]")
    N, (H, W) = centres.shape[0], depth.shape[1:]
    J = centres.shape[1]
    if radii.numel() != J:
        raise RuntimeError("radii must have J entries")
    if depth_index is not None:
        _check_index(depth_index, N, depth.shape[0], "depth_index")
    if N > 0 and d2m_two_step_pays(depth, shared=depth_index is not None and N >= 2 * depth.shape[0]):
        # compact every image once, search the point lists (the same integer sums, see csrc/data_to_model.hip)
        ws = d2m_compact(depth)
        return data_to_model_from_points(ws, depth.shape[0], H, W, centres, radii, depth_index, want_grad)
    lib = _lib.lib()
    R = lib.shr_data_to_model_parts(N, int(H), int(W))     # large crops: R partial results per crop, added here
    with _on(depth.device):
        loss_sum = torch.empty((N, R), dtype=torch.float32, device=depth.device)
        grad = torch.empty((N, R, J, 3), dtype=torch.float32, device=depth.device) if want_grad else None
        _lib.check(lib.shr_data_to_model_partial(_ptr(depth), _ptr(depth_index), _ptr(centres), 3, _ptr(radii), N, J, H, W, R,
                                                 _ptr(loss_sum), _ptr(grad), _stream()), "shr_data_to_model_partial")
        if R > 1:
            loss_sum = loss_sum.sum(1)
            grad = grad.sum(1) if want_grad else None
        else:
            loss_sum = loss_sum.view(N)
            grad = grad.view(N, J, 3) if want_grad else None
    return (loss_sum, grad) if want_grad else loss_sum


D2M_TWO_STEP_MIN_PIXELS = 1 << 23   # (384 images @128x128 = 6.3 M pixels: the two paths tie there -- 52 vs 55 us at 1152 crops)


def d2m_points_supported(depth):
    """True when the two-step data->model path takes this image stack [M,H,W] (16-byte rows, at most 2^28 pixels)."""
    return depth.dim() == 3 and depth.shape[0] > 0 and depth.data_ptr() % 16 == 0 and \
        _lib.lib().shr_data_to_model_points_bytes(int(depth.shape[0]), int(depth.shape[1]), int(depth.shape[2])) > 0


def d2m_two_step_pays(depth, shared=False):
    """The two-step path costs a launch more than the streaming kernel and wins where the streaming kernel is bound by
    its VALU work (large images compared with several sphere sets); small stacks stay with one launch.  shared: the
    stand-alone term with every image searched by at least two crops -- the compaction is paid once per image, the
    search per crop: from half the size on (1152 crops on 384 images @128x128: 46 against 56 us; the fused loss, which
    has other launches to feed, gains nothing there and keeps the full threshold)."""
    least = D2M_TWO_STEP_MIN_PIXELS // 2 if shared else D2M_TWO_STEP_MIN_PIXELS
    return D2M_TWO_STEP and depth.numel() >= least and d2m_points_supported(depth)


MV_OVERLAP = True        # MutualProjectionLossFused: render-and-compare beside the point search (see there)
SAME_VIEW_SPLIT = True   # ... and, for the same-view pairs only, the compare on those pairs alone (see there)
# The side stream and its 'projection done' event are kept per (device, caller's stream): two callers on different
# streams of one device (two training loops in one process) get a side stream and an event each, so neither orders the
# other's work.  One caller stream = one pair, reused call after call (creating them costs ~20 us).
_SIDE = {}


def _side_key(dev):
    d = dev.index if dev.index is not None else _cur_device()
    return (d, _raw_stream(d))


def _side_stream(dev):
    k = _side_key(dev)
    if k not in _SIDE:
        _SIDE[k] = torch.cuda.Stream(device=dev)
    return _SIDE[k]


_SIDE_EVENT = {}


def _side_event(dev):
    """One reusable event per (device, caller's stream) for the 'projection done' edge between the two streams."""
    k = _side_key(dev)
    if k not in _SIDE_EVENT:
        _SIDE_EVENT[k] = torch.cuda.Event()
    return _SIDE_EVENT[k]


def d2m_points_workspace(depth):
    """An UNFILLED workspace (uint8 tensor) for the point lists of depth [M,H,W] (shr_data_to_model_points_bytes)."""
    _check_input(depth, "depth")
    M, H, W = depth.shape
    nbytes = _lib.lib().shr_data_to_model_points_bytes(int(M), int(H), int(W))
    if nbytes <= 0 or depth.data_ptr() % 16:
        raise RuntimeError("image not taken by the two-step data->model path (rows of 4 pixels, 16-byte aligned)")
    with _on(depth.device):
        return torch.empty(nbytes, dtype=torch.uint8, device=depth.device)


def d2m_compact(depth):
    """depth [M,H,W] -> workspace (uint8 tensor): every image's foreground pixels as tile-sorted point records, for
    data_to_model_from_points (shr_data_to_model_compact)."""
    ws = d2m_points_workspace(depth)
    M, H, W = depth.shape
    with _on(depth.device):
        _lib.check(_lib.lib().shr_data_to_model_compact(_ptr(depth), M, H, W, _ptr(ws), _stream()), "shr_data_to_model_compact")
    return ws


def d2m_points_parts(N):
    """Partial results per crop of the point-list search: ONE.  The regions of an image enter its list in the order
    their workgroups arrive, so the groups a part would own change from run to run; a crop's total is an integer sum
    over all its points and does not (bit-reproducible), several float partials added afterwards would only be equal
    to a rounding.  The launcher gives a crop's workgroup more waves when there are few crops."""
    return 1


def data_to_model_from_points(ws, M, H, W, centres, radii, depth_index=None, want_grad=False, centre_stride=None, parts=None):
    """The search step: centres [N,J,3] (or the rasterizer's [N,J,4] records with centre_stride=4) against the point
    lists of images depth_index[n] (or n) -> loss_sum [N] (and grad_centres [N,J,3])."""
    N, J = centres.shape[0], centres.shape[1]
    stride = int(centres.shape[2]) if centre_stride is None else centre_stride
    P = d2m_points_parts(N) if parts is None else parts
    lib = _lib.lib()
    with _on(centres.device):
        loss = torch.empty((N, P), dtype=torch.float32, device=centres.device)
        grad = torch.empty((N, P, J, 3), dtype=torch.float32, device=centres.device) if want_grad else None
        _lib.check(lib.shr_data_to_model_from_points(_ptr(ws), int(M), _ptr(depth_index), _ptr(centres), stride, _ptr(radii), N, J,
                                                     int(H), int(W), P, _ptr(loss), _ptr(grad), _stream()),
                   "shr_data_to_model_from_points")
        if P > 1:
            loss = loss.sum(1)
            grad = grad.sum(1) if want_grad else None
        else:
            loss = loss.view(N)
            grad = grad.view(N, J, 3) if want_grad else None
    return (loss, grad) if want_grad else loss


class DataToModel(torch.autograd.Function):
    """mean over ALL N*H*W pixels of the clamped point-to-sphere-surface distance
    (mesh/render.py:123-142).  The kernel emits the unit gradient with the loss;
    backward only scales it."""

    @staticmethod
    def forward(ctx, depth, centres, radii, depth_index=None):
        depth = depth.contiguous()
        centres = centres.contiguous()
        count = float(centres.shape[0] * depth.shape[-2] * depth.shape[-1])   # pixels of the N crops
        if ctx.needs_input_grad[1]:
            loss_sum, grad = data_to_model(depth, centres, radii, want_grad=True, depth_index=depth_index)
            ctx.save_for_backward(grad)
            ctx.count = count
        else:
            loss_sum = data_to_model(depth, centres, radii, depth_index=depth_index)
        return (loss_sum.double().sum() / count).float()

    @staticmethod
    def backward(ctx, grad_out):
        (grad,) = ctx.saved_tensors
        return None, grad * (grad_out / ctx.count), None, None


class MutualProjectionLossFused(torch.autograd.Function):
    """(cam, inv_cam [B,V,4,4], joints [B,V,J,3], observed [B*V,H,W], radii [J], index [B*V*V] int32 (pair -> observed
    image), diag_index [B*V] int32 (the same-view pairs' numbers), is_mv, ..., diag_target [B*V] int32 = index[diag_index]) ->
    (loss, projected depth [B*V*V,H,W]): MutualProjectionLoss (mesh/multiview_utility.py:90-130) as FIVE launches
    -- view projection, (compaction of the observed images into point lists,) fused render-and-compare, data->model,
    and the assembly kernel that weights, adds and pulls both sphere gradients back to the joints (the whole backward
    is done in the forward: the losses are plain sums) -- plus one scaling in backward.  The unfused wiring needed
    ~35 small torch launches around the same three kernels.
    points_ws: a point-list workspace of these very observed images (d2m_points_workspace); points_fresh = it is
    still to be filled (done here, by the call that also projects the views: shr_mv_project_compact)."""

    @staticmethod
    def forward(ctx, cam, inv_cam, joints, observed, radii, index, diag_index, is_mv, d2m_weight, points_ws=None,
                points_fresh=False, diag_target=None, want_depth=True, order=None, order_target=None):
        cam, inv_cam, joints = (t.detach().contiguous().float() for t in (cam, inv_cam, joints))
        observed, radii = observed.contiguous().float(), radii.contiguous().float()
        for t, name in ((cam, "camera_poses"), (inv_cam, "inv_camera_poses"), (joints, "joints"), (observed, "depth_maps"),
                        (radii, "radii")):
            _check_input(t, name)
        B, V, J = joints.shape[0], joints.shape[1], joints.shape[2]
        H, W = observed.shape[-2], observed.shape[-1]
        N = B * V * V
        lib = _lib.lib()
        if diag_index.dtype != torch.int32:
            diag_index = diag_index.to(torch.int32)
        if diag_target is None and not is_mv:
            diag_target = index.index_select(0, diag_index.long())
        Rm = lib.shr_sphere_raster_mse_regions(int(H), int(W))
        dev = joints.device
        if N == 0 or J == 0:     # (the entry points return at once on an empty batch: nothing would write the outputs)
            if ctx.needs_input_grad[2]:
                ctx.save_for_backward(torch.zeros((B, V, J, 3), dtype=torch.float32, device=dev))
            depth = torch.full((N, H, W), 100.0, dtype=torch.float32, device=dev)
            ctx.mark_non_differentiable(depth)
            ctx.set_materialize_grads(False)
            return torch.zeros((), dtype=torch.float32, device=dev), depth
        # every observed image is compared with V sphere sets (mesh/multiview_utility.py:99): compacted once into a
        # point list (points_ws: the caller's -- MutualProjectionLoss keeps the lists while it is handed the same
        # observations again: a second hourglass stack, a fitting loop)
        two_step = points_ws is not None or d2m_two_step_pays(observed)
        if two_step and points_ws is None:
            points_ws, points_fresh = d2m_points_workspace(observed), True
        with _on(dev):
            spheres = torch.empty((N, J, 4), dtype=torch.float32, device=dev)
            # want_depth = False (MutualProjectionLoss.return_projections off: a training loop that never looks at the
            # projections): the fused kernel writes no depth map -- half of its HBM bytes -- and the same-view mode
            # launches no plain forward at all; an empty tensor stands in for the projections.
            depth = torch.empty((N, H, W) if want_depth else (0, H, W), dtype=torch.float32, device=dev)
            dptr = _ptr(depth) if want_depth else None
            # Two streams where the stack is large enough for the two-step path (MV_OVERLAP; same bits either way):
            #   caller's stream   view projection -> render-and-compare (the LONG kernel: nothing it waits for crosses
            #                     a queue) [-> the plain forward of all pairs in the same-view split]
            #   side stream       compaction of the observed images (needs nothing from this call) -> [projection
            #                     done] -> point search
            # Round 4 had projection + compaction on the caller's stream and the render-and-compare kernel on the
            # side: it started 12 us after the compaction's end (cross-queue dependency) and the join cost another 11
            # on the critical path (rocprofv3 timeline, tools/timeline_mvloss.sh).
            overlap = MV_OVERLAP and two_step
            main = side = None
            s_main = _stream()
            s_d2m = s_main
            if overlap:
                main, side = torch.cuda.current_stream(dev), _side_stream(dev)
                side.wait_stream(main)
                s_d2m = side.cuda_stream
            if two_step and points_fresh and not overlap:
                _lib.check(lib.shr_mv_project_compact(_ptr(cam), _ptr(inv_cam), _ptr(joints), _ptr(radii), B, V, J,
                                                      _ptr(spheres), _ptr(observed), int(observed.shape[0]), H, W,
                                                      _ptr(points_ws), s_main), "shr_mv_project_compact")
            else:
                if two_step and points_fresh:
                    _lib.check(lib.shr_data_to_model_compact(_ptr(observed), int(observed.shape[0]), H, W, _ptr(points_ws), s_d2m),
                               "shr_data_to_model_compact")
                _lib.check(lib.shr_mutual_project_fwd(_ptr(cam), _ptr(inv_cam), _ptr(joints), _ptr(radii), B, V, J,
                                                      _ptr(spheres), s_main), "shr_mutual_project_fwd")
            if overlap:
                ev = _side_event(dev)
                ev.record(main)
                side.wait_event(ev)               # the point search reads the projected records
            # Same-view pairs only (what the reference trains with after its first 1500 iterations,
            # network/engine.py:361): E = B*V pairs enter the loss, all V*V projections are still returned.  The pairs
            # are SELECTED by index inside the kernels (diag_index: pair e = crop diag_index[e] of the batch; diag_target:
            # its observed image) -- no gather of their records in front (two torch launches and 16 us until round 4).
            if is_mv:
                E, cidx, cen_index = N, index, None
            else:
                E, cidx, cen_index = B * V, diag_target, diag_index
            # On a large stack the compare runs on those pairs alone (no depth output) and the plain forward kernel
            # renders all N projections behind it (SAME_VIEW_SPLIT).
            split = (overlap or not want_depth) and not is_mv and SAME_VIEW_SPLIT
            Em = E if split else N
            # the partial results of the two terms: ONE allocation, addressed by offsets (four torch.empty calls and
            # their tensors were ~10 us of host time per step)
            Rd = d2m_points_parts(E) if two_step else lib.shr_data_to_model_parts(E, int(H), int(W))
            n_gsp, n_gd2m, n_sse, n_d2m = Em * Rm * J * 4, E * Rd * J * 3, Em * Rm, E * Rd
            scratch = torch.empty(n_gsp + n_gd2m + n_sse + n_d2m, dtype=torch.float32, device=dev)
            loss = torch.empty(1, dtype=torch.float32, device=dev)    # (its own: the result must not pin the scratch)
            p_gsp = scratch.data_ptr()                          # (16-byte aligned: the allocation's start)
            p_gd2m = p_gsp + 4 * n_gsp
            p_sse = p_gd2m + 4 * n_gd2m
            p_d2m = p_sse + 4 * n_sse
            if split:
                _lib.check(lib.shr_sphere_raster_mse_indexed(_ptr(spheres), _ptr(diag_index), E, J, H, W, _ptr(observed),
                                                             _ptr(index), None, p_sse, p_gsp, s_main),
                           "shr_sphere_raster_mse_indexed")
                if want_depth:
                    _lib.check(lib.shr_sphere_raster_fwd_ex(_ptr(spheres), N, J, H, W, dptr, None, 0, s_main),
                               "shr_sphere_raster_fwd_ex")
            elif is_mv and order is not None:
                # all pairs, in the XCD-aware launch order (multiview_utility._indices): the same slots, the same bits
                _lib.check(lib.shr_sphere_raster_mse_ordered(_ptr(spheres), _ptr(order), N, J, H, W, _ptr(observed), _ptr(index),
                                                             dptr, p_sse, p_gsp, s_main), "shr_sphere_raster_mse_ordered")
            else:
                _lib.check(lib.shr_sphere_raster_mse(_ptr(spheres), N, J, H, W, _ptr(observed), _ptr(index), dptr,
                                                     p_sse, p_gsp, s_main), "shr_sphere_raster_mse")
            if two_step:
                ws = points_ws
                # (ordered: workgroup = crop; with several parts per crop blockIdx.x = crop * parts + part and the XCD
                # placement the order was built for no longer holds -- the batch's own order then)
                if is_mv and order is not None and Rd == 1:
                    _lib.check(lib.shr_data_to_model_from_points_ordered(_ptr(ws), int(observed.shape[0]), _ptr(order_target),
                                                                         _ptr(order), _ptr(spheres), 4, _ptr(radii), E, J, H, W,
                                                                         Rd, p_d2m, p_gd2m, s_d2m), "shr_data_to_model_from_points_ordered")
                else:
                    _lib.check(lib.shr_data_to_model_from_points_indexed(_ptr(ws), int(observed.shape[0]), _ptr(cidx), _ptr(cen_index),
                                                                         _ptr(spheres), 4, _ptr(radii), E, J, H, W, Rd, p_d2m,
                                                                         p_gd2m, s_d2m), "shr_data_to_model_from_points")
                if overlap:
                    # the join: everything the side stream wrote into `scratch` / read from `spheres` is ordered before
                    # the assembly kernel on the caller's stream -- and before the caching allocator can hand either
                    # block to a later allocation of that stream (which is why no record_stream() is needed)
                    main.wait_stream(side)
            else:
                cen = spheres if is_mv else spheres.index_select(0, diag_index.long())
                _lib.check(lib.shr_data_to_model_partial(_ptr(observed), _ptr(cidx), _ptr(cen), 4, _ptr(radii), E, J, H, W, Rd,
                                                         p_d2m, p_gd2m, s_main), "shr_data_to_model_partial")
            want = ctx.needs_input_grad[2]
            gj = torch.empty((B, V, J, 3), dtype=torch.float32, device=dev) if want else None
            _lib.check(lib.shr_mv_loss_combine(_ptr(cam), _ptr(inv_cam), p_sse, p_gsp, Rm, p_d2m, p_gd2m, Rd,
                                               B, V, J, H, W, 1 if is_mv else (2 if split else 0), float(d2m_weight), _ptr(loss), _ptr(gj),
                                               _stream()), "shr_mv_loss_combine")
        if want:
            ctx.save_for_backward(gj)
        ctx.mark_non_differentiable(depth)
        ctx.set_materialize_grads(False)
        return loss.view(()), depth

    @staticmethod
    def backward(ctx, g_loss, _g_depth):
        if g_loss is None:
            return (None,) * 15
        (gj,) = ctx.saved_tensors
        return (None, None, gj * g_loss) + (None,) * 12


class MutualProject(torch.autograd.Function):
    """(cam, inv_cam [B,V,4,4], joints [B,V,J,3], radii [J]) -> spheres [B,V,V,J,4]:
    every view's joints expressed in every view (mesh/multiview_utility.py:13-30,
    :62-72).  Differentiable w.r.t. joints only (the transforms are detached in
    the reference, :68)."""

    @staticmethod
    def forward(ctx, cam, inv_cam, joints, radii):
        cam, inv_cam, joints, radii = (t.contiguous().float() for t in (cam, inv_cam, joints, radii))
        for t, name in ((cam, "camera_poses"), (inv_cam, "inv_camera_poses"), (joints, "joints"), (radii, "radii")):
            _check_input(t, name)
        B, V, J = joints.shape[0], joints.shape[1], joints.shape[2]
        if cam.shape != (B, V, 4, 4) or inv_cam.shape != (B, V, 4, 4) or joints.shape[3] != 3:
            raise RuntimeError("expected cam/inv_cam [B,V,4,4] and joints [B,V,J,3]")
        with _on(joints.device):
            out = torch.empty((B, V, V, J, 4), dtype=torch.float32, device=joints.device)
            _lib.check(_lib.lib().shr_mutual_project_fwd(_ptr(cam), _ptr(inv_cam), _ptr(joints), _ptr(radii), B, V, J,
                                                         _ptr(out), _stream()), "shr_mutual_project_fwd")
        ctx.save_for_backward(cam, inv_cam)
        return out

    @staticmethod
    def backward(ctx, grad_spheres):
        cam, inv_cam = ctx.saved_tensors
        g = grad_spheres.contiguous()
        B, V, _, J, _ = g.shape
        with _on(g.device):
            out = torch.empty((B, V, J, 3), dtype=torch.float32, device=g.device)
            _lib.check(_lib.lib().shr_mutual_project_bwd(_ptr(cam), _ptr(inv_cam), _ptr(g), B, V, J, _ptr(out),
                                                         _stream()), "shr_mutual_project_bwd")
        return None, None, out, None


def tri_raster_fwd(width, height, face_vertices):
    """face_vertices [B,F,3,3] (pixel-space x,y,z) -> depth [B,height,width],
    background 1000: depth_rasterization.forward (mesh/cuda_kernel)."""
    _check_input(face_vertices, "vertices")
    if face_vertices.dim() < 2:
        raise RuntimeError("vertices must be [B,F,3,3]")
    B, F = face_vertices.shape[0], face_vertices.shape[1]
    if face_vertices.numel() != B * F * 9:
        raise RuntimeError("vertices must hold 9 floats per face ([B,F,3,3])")
    with _on(face_vertices.device):
        depth = torch.empty((B, height, width), dtype=torch.float32, device=face_vertices.device)
        _lib.check(_lib.lib().shr_tri_raster_fwd(_ptr(face_vertices), B, F, width, height, _ptr(depth), _stream()),
                   "shr_tri_raster_fwd")
    return depth


def tri_raster_indexed_fwd(width, height, vertices, faces):
    """vertices [B,NV,4] + faces [F,3] int32 -> depth [B,height,width] (gather fused)."""
    _check_input(vertices, "vertices")
    _check_input(faces, "faces", torch.int32)
    if vertices.dim() != 3 or vertices.shape[2] != 4 or faces.dim() != 2 or faces.shape[1] != 3:
        raise RuntimeError("vertices must be [B,NV,4] and faces [F,3]")
    B, NV = vertices.shape[0], vertices.shape[1]
    with _on(vertices.device):
        depth = torch.empty((B, height, width), dtype=torch.float32, device=vertices.device)
        _lib.check(_lib.lib().shr_tri_raster_indexed_fwd(_ptr(vertices), _ptr(faces), B, NV, faces.shape[0], width,
                                                         height, _ptr(depth), _stream()),
                   "shr_tri_raster_indexed_fwd")
    return depth


def lbs_project(T, skin_vertex_start, skin_bone, skin_wv, right_hand=True, camera=None, rand_f=None):
    """T [B,NB,4,4] -> skinned (and, with camera=(cx,cy,fx,fy), projected) vertices [B,NV,4]."""
    _check_input(T, "bone_transformations")
    _check_input(skin_vertex_start, "skin_vertex_start", torch.int32)
    _check_input(skin_bone, "skin_bone", torch.int32)
    _check_input(skin_wv, "skin_wv")
    if rand_f is not None:
        _check_input(rand_f, "rand_f")
    B, NB = T.shape[0], T.shape[1]
    NV = skin_vertex_start.numel() - 1
    cx, cy, fx, fy = camera if camera is not None else (0.0, 0.0, 1.0, 1.0)
    with _on(T.device):
        out = torch.empty((B, NV, 4), dtype=torch.float32, device=T.device)
        _lib.check(_lib.lib().shr_lbs_project(_ptr(T), B, NB, NV, _ptr(skin_vertex_start), _ptr(skin_bone),
                                              _ptr(skin_wv), int(bool(right_hand)), int(camera is not None),
                                              cx, cy, fx, fy, _ptr(rand_f), _ptr(out), _stream()), "shr_lbs_project")
    return out


class KeypointSpheres(torch.autograd.Function):
    """T[B,NB,4,4] -> the rasterizer's records spheres[B,J,4] = (+-p.x, p.y, p.z, radii[j]), p = T[bone[j]] @ wv[j]: the
    key-point skinning + cat of HandBallPrimitiveRender (mesh/render.py:65-88) as one launch per direction; gradient
    w.r.t. T (the radii are buffers in the reference).  Tables: pointTransformation.LinearBlendSkinning.kp_*."""

    @staticmethod
    def forward(ctx, T, bone, wv, radii, bone_start, bone_points, right_hand):
        T = T.contiguous().float()
        _check_input(T, "transformation matrices")
        if T.dim() != 4 or T.shape[2:] != (4, 4):
            raise RuntimeError("transformation matrices must be [B,NB,4,4]")
        B, NB, J = T.shape[0], T.shape[1], bone.shape[0]
        if bone_start.shape[0] != NB + 1 or wv.shape != (J, 4) or radii.numel() != J:
            raise RuntimeError("key-point tables do not match the bones")
        with _on(T.device):
            sph = torch.empty((B, J, 4), dtype=torch.float32, device=T.device)
            _lib.check(_lib.lib().shr_keypoint_spheres_fwd(_ptr(T), B, NB, J, _ptr(bone), _ptr(wv), _ptr(radii),
                                                           int(bool(right_hand)), _ptr(sph), _stream()),
                       "shr_keypoint_spheres_fwd")
        ctx.save_for_backward(wv, bone_start, bone_points)
        ctx.dims = (B, NB, J, int(bool(right_hand)))
        return sph

    @staticmethod
    def backward(ctx, grad_spheres):
        wv, bone_start, bone_points = ctx.saved_tensors
        B, NB, J, right = ctx.dims
        g = grad_spheres.contiguous().float()
        with _on(g.device):
            gT = torch.zeros((B, NB, 4, 4), dtype=torch.float32, device=g.device) if B == 0 else \
                torch.empty((B, NB, 4, 4), dtype=torch.float32, device=g.device)
            _lib.check(_lib.lib().shr_keypoint_spheres_bwd(_ptr(g), B, NB, J, _ptr(bone_start), _ptr(bone_points),
                                                           _ptr(wv), right, _ptr(gT), _stream()),
                       "shr_keypoint_spheres_bwd")
        return gT, None, None, None, None, None, None


class ForwardKinematics(torch.autograd.Function):
    """params [B,26] -> bone transforms [B,17,4,4] (mesh/kinematicsTransformation.py:169-177),
    analytic backward; offset / offset_inv [17,4,4] are constants."""

    @staticmethod
    def forward(ctx, params, offset, offset_inv):
        params = params.contiguous().float()
        for t, name in ((params, "parameters"), (offset, "offset"), (offset_inv, "offset_inv")):
            _check_input(t, name)
        if params.dim() != 2 or params.shape[1] != 26:
            raise RuntimeError("parameters must be [B,26]")
        B = params.shape[0]
        with _on(params.device):
            T = torch.empty((B, 17, 4, 4), dtype=torch.float32, device=params.device)
            _lib.check(_lib.lib().shr_fk_fwd(_ptr(params), B, _ptr(offset), _ptr(offset_inv), _ptr(T), _stream()),
                       "shr_fk_fwd")
        ctx.save_for_backward(params, offset, offset_inv)
        return T

    @staticmethod
    def backward(ctx, grad_T):
        params, offset, offset_inv = ctx.saved_tensors
        g = grad_T.contiguous().float()
        B = params.shape[0]
        with _on(params.device):
            out = torch.empty((B, 26), dtype=torch.float32, device=params.device)
            _lib.check(_lib.lib().shr_fk_bwd(_ptr(params), B, _ptr(offset), _ptr(offset_inv), _ptr(g), _ptr(out),
                                             _stream()), "shr_fk_bwd")
        return out, None, None


class PoseSpheres(torch.autograd.Function):
    """params [B,26] -> the rasterizer's records spheres [B,J,4]: ForwardKinematics followed by KeypointSpheres as ONE
    launch per direction (shr_pose_spheres_fwd / _bwd): the bone transforms stay in LDS.  The fit chain
    HandBallPrimitiveRender(HandTransformationMat(pose)) (mesh/render.py:81-88 over
    mesh/kinematicsTransformation.py:169-177); records and gradient are bit-identical to the two Functions chained."""

    @staticmethod
    def forward(ctx, params, offset, offset_inv, bone, wv, radii, bone_start, bone_points, right_hand):
        params = params.contiguous().float()
        for t, name in ((params, "parameters"), (offset, "offset"), (offset_inv, "offset_inv"), (wv, "key-points"),
                        (radii, "radii")):
            _check_input(t, name)
        if params.dim() != 2 or params.shape[1] != 26:
            raise RuntimeError("parameters must be [B,26]")
        B, J = params.shape[0], bone.shape[0]
        if bone_start.shape[0] != 18 or wv.shape != (J, 4) or radii.numel() != J:
            raise RuntimeError("key-point tables do not match the 17 bones")
        with _on(params.device):
            sph = torch.empty((B, J, 4), dtype=torch.float32, device=params.device)
            _lib.check(_lib.lib().shr_pose_spheres_fwd(_ptr(params), B, _ptr(offset), _ptr(offset_inv), J, _ptr(bone), _ptr(wv),
                                                       _ptr(radii), int(bool(right_hand)), _ptr(sph), None, _stream()),
                       "shr_pose_spheres_fwd")
        ctx.save_for_backward(params, offset, offset_inv, wv, bone_start, bone_points)
        ctx.dims = (B, J, int(bool(right_hand)))
        return sph

    @staticmethod
    def backward(ctx, grad_spheres):
        params, offset, offset_inv, wv, bone_start, bone_points = ctx.saved_tensors
        B, J, right = ctx.dims
        g = grad_spheres.contiguous().float()
        with _on(g.device):
            out = torch.empty((B, 26), dtype=torch.float32, device=g.device)
            _lib.check(_lib.lib().shr_pose_spheres_bwd(_ptr(params), B, _ptr(offset), _ptr(offset_inv), J, _ptr(bone_start),
                                                       _ptr(bone_points), _ptr(wv), right, _ptr(g), _ptr(out), _stream()),
                       "shr_pose_spheres_bwd")
        return out, None, None, None, None, None, None, None, None


class PoseDepthRaster(torch.autograd.Function):
    """params [B,26] -> depth [B,H,W]: PoseSpheres followed by SphereDepthRaster as ONE autograd node (two launches per
    direction; an eager fitting loop pays the Python / autograd cost of one Function instead of two).  Same kernels,
    same bits as the two Functions chained."""

    @staticmethod
    def forward(ctx, params, offset, offset_inv, bone, wv, radii, bone_start, bone_points, right_hand, H, W):
        params = params.contiguous().float()
        _check_input(params, "parameters")
        if params.dim() != 2 or params.shape[1] != 26:
            raise RuntimeError("parameters must be [B,26]")
        B, J = params.shape[0], bone.shape[0]
        if bone_start.shape[0] != 18 or wv.shape != (J, 4) or radii.numel() != J:
            raise RuntimeError("key-point tables do not match the 17 bones")
        lib, right = _lib.lib(), int(bool(right_hand))
        want = ctx.needs_input_grad[0]
        with _on(params.device):
            s = _stream()
            sph = torch.empty((B, J, 4), dtype=torch.float32, device=params.device)
            depth = torch.empty((B, H, W), dtype=torch.float32, device=params.device)
            owner = torch.empty((B, H, W), dtype=torch.uint8, device=params.device) if want else None
            _lib.check(lib.shr_pose_spheres_fwd(_ptr(params), B, _ptr(offset), _ptr(offset_inv), J, _ptr(bone), _ptr(wv),
                                                _ptr(radii), right, _ptr(sph), None, s), "shr_pose_spheres_fwd")
            _lib.check(lib.shr_sphere_raster_fwd_ex(_ptr(sph), B, J, H, W, _ptr(depth), _ptr(owner),
                                                    RASTER_OWNER_TOUCHED_ROWS if want else 0, s), "shr_sphere_raster_fwd")
        if want:
            ctx.save_for_backward(params, offset, offset_inv, wv, bone_start, bone_points, sph, owner)
            ctx.dims = (B, J, right, H, W)
        return depth

    @staticmethod
    def backward(ctx, grad_depth):
        params, offset, offset_inv, wv, bone_start, bone_points, sph, owner = ctx.saved_tensors
        B, J, right, H, W = ctx.dims
        g = grad_depth.contiguous()
        lib = _lib.lib()
        with _on(g.device):
            s = _stream()
            gs = torch.empty((B, J, 4), dtype=torch.float32, device=g.device)
            out = torch.empty((B, 26), dtype=torch.float32, device=g.device)
            _lib.check(lib.shr_sphere_raster_bwd(_ptr(sph), _ptr(g), _ptr(owner), B, J, H, W, _ptr(gs), s), "shr_sphere_raster_bwd")
            _lib.check(lib.shr_pose_spheres_bwd(_ptr(params), B, _ptr(offset), _ptr(offset_inv), J, _ptr(bone_start),
                                                _ptr(bone_points), _ptr(wv), right, _ptr(gs), _ptr(out), s), "shr_pose_spheres_bwd")
        return (out,) + (None,) * 10


def mesh_depth_fwd(vertices, faces, out_size, src_size=640, clamp_max=100.0):
    """vertices [B,NV,4] (src_size pixel space) + faces [F,3] int32 -> depth [B,S,S]: triangle
    raster at src_size, clamp(max), bilinear resize to S, fused (only the sampled pixels)."""
    _check_input(vertices, "vertices")
    _check_input(faces, "faces", torch.int32)
    if vertices.dim() != 3 or vertices.shape[2] != 4 or faces.dim() != 2 or faces.shape[1] != 3:
        raise RuntimeError("vertices must be [B,NV,4] and faces [F,3]")
    B, NV = vertices.shape[0], vertices.shape[1]
    with _on(vertices.device):
        depth = torch.empty((B, out_size, out_size), dtype=torch.float32, device=vertices.device)
        _lib.check(_lib.lib().shr_mesh_depth_fwd(_ptr(vertices), _ptr(faces), B, NV, faces.shape[0], src_size, out_size,
                                                 clamp_max, _ptr(depth), _stream()), "shr_mesh_depth_fwd")
    return depth


def mesh_render_fwd(T, skin_vertex_start, skin_bone, skin_wv, right_hand, camera, rand_f, faces, out_size, src_size=640,
                    clamp_max=100.0):
    """DepthRender.forward as one call (shr_mesh_render_fwd): T [B,NB,4,4] -> depth [B,S,S].  One launch where the fused
    kernel applies (integer src_size / S, lattice <= 128 x 128); otherwise skinning and raster through a vertex workspace."""
    _check_input(T, "bone_transformations")
    _check_input(skin_vertex_start, "skin_vertex_start", torch.int32)
    _check_input(skin_bone, "skin_bone", torch.int32)
    _check_input(skin_wv, "skin_wv")
    _check_input(faces, "faces", torch.int32)
    if rand_f is not None:
        _check_input(rand_f, "rand_f")
    if T.dim() != 4 or T.shape[2:] != (4, 4) or faces.dim() != 2 or faces.shape[1] != 3:
        raise RuntimeError("T must be [B,NB,4,4] and faces [F,3]")
    B, NB = T.shape[0], T.shape[1]
    NV = skin_vertex_start.numel() - 1
    cx, cy, fx, fy = camera
    with _on(T.device):
        depth = torch.empty((B, out_size, out_size), dtype=torch.float32, device=T.device)
        ws = torch.empty((B, NV, 4), dtype=torch.float32, device=T.device)     # (the caching allocator: no launch; unused
        #                                                                         when the fused kernel takes the call)
        _lib.check(_lib.lib().shr_mesh_render_fwd(_ptr(T), B, NB, NV, _ptr(skin_vertex_start), _ptr(skin_bone), _ptr(skin_wv),
                                                  int(bool(right_hand)), cx, cy, fx, fy, _ptr(rand_f), _ptr(faces),
                                                  faces.shape[0], src_size, out_size, clamp_max, _ptr(ws), _ptr(depth),
                                                  _stream()), "shr_mesh_render_fwd")
    return depth


def hand_synth(params, offset, offset_inv, rng_state, rand_scale, lbs, faces, camera, out_size, depth_scale, noise,
               sigma_xy, sigma_z, heat=None, src_size=640, clamp_max=100.0):
    """HandSynthesizer.forward in ONE launch (shr_hand_synth_fwd), or None where that kernel does not apply (the caller
    then takes synth_pose / mesh_render_post / heatmap_render).  lbs: the mesh's SparseSkinning (tables + right_hand);
    heat: None (depth only) or (kp_start, kp_bone_i32, kp_wv, hm, (hcx, hcy, hfx, hfy), sigma, inv_k, uv_scale, d_scale).
    Returns (draws [6,B], depth [B,S,S], uv_hm, d_hm, xyz) -- the last three None without `heat`."""
    params = params.contiguous().float()
    B = params.shape[0]
    NV, F = lbs.skin_vertex_start.numel() - 1, faces.shape[0]
    J, hm = (heat[0].numel() - 1, int(heat[3])) if heat is not None else (0, 1)
    lib = _lib.lib()
    if params.dim() != 2 or params.shape[1] != 26 or not lib.shr_hand_synth_one_launch(17, NV, F, src_size, out_size, J, hm):
        return None
    _check_input(params, "parameters")
    _check_input(rng_state, "rng_state", torch.int64)
    dev = params.device
    cx, cy, fx, fy = camera
    with _on(dev):
        draws = torch.empty((6, B), dtype=torch.float32, device=dev)
        depth = torch.empty((B, out_size, out_size), dtype=torch.float32, device=dev)
        uv = d = xyz = None
        hargs = (0, None, None, None, 1, 0.0, 0.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0, 1.0, 0.0)
        if heat is not None:
            kp_start, kp_bone, kp_wv, _, hcam, hsigma, inv_k, uv_scale, d_scale = heat
            uv = torch.empty((B, J, hm, hm), dtype=torch.float32, device=dev)
            d = torch.empty((B, J, hm, hm), dtype=torch.float32, device=dev)
            xyz = torch.empty((B, J, 4), dtype=torch.float32, device=dev)
            hargs = (J, _ptr(kp_start), _ptr(kp_bone), _ptr(kp_wv), hm, hcam[0], hcam[1], hcam[2], hcam[3], float(hsigma),
                     float(uv_scale), float(d_scale), inv_k[0], inv_k[1], inv_k[2], inv_k[3])
        _lib.check(lib.shr_hand_synth_fwd(_ptr(params), B, _ptr(offset), _ptr(offset_inv), _ptr(rng_state), float(rand_scale), NV,
                                          _ptr(lbs.skin_vertex_start), _ptr(lbs.skin_bone), _ptr(lbs.skin_wv),
                                          int(bool(lbs.right_hand)), cx, cy, fx, fy, _ptr(faces), F, src_size, out_size,
                                          clamp_max, float(depth_scale), int(bool(noise)), float(sigma_xy), float(sigma_z),
                                          *hargs, _ptr(draws), _ptr(depth), _ptr(uv), _ptr(d), _ptr(xyz), _stream()),
                   "shr_hand_synth_fwd")
    return draws, depth, uv, d, xyz


def synth_pose(params, offset, offset_inv, rng_state, rand_scale):
    """pose [B,26] -> (diag(s) T [B,17,4,4], draws [6,B]): forward kinematics, RandScale and the samples' random draws in
    one launch (shr_synth_pose_fwd).  draws = s_x, s_y, s_z, focal jitter, and the two uint32 noise keys (as float bits);
    rng_state: int64 [2] = (seed, call counter) on the device (read only here)."""
    params = params.contiguous().float()
    for t, name in ((params, "parameters"), (offset, "offset"), (offset_inv, "offset_inv")):
        _check_input(t, name)
    _check_input(rng_state, "rng_state", torch.int64)
    if params.dim() != 2 or params.shape[1] != 26 or rng_state.numel() < 3:
        raise RuntimeError("parameters must be [B,26] and rng_state int64 [>= 3] (seed, call counter, ticket)")
    B = params.shape[0]
    with _on(params.device):
        T = torch.empty((B, 17, 4, 4), dtype=torch.float32, device=params.device)
        draws = torch.empty((6, B), dtype=torch.float32, device=params.device)
        _lib.check(_lib.lib().shr_synth_pose_fwd(_ptr(params), B, _ptr(offset), _ptr(offset_inv), _ptr(rng_state),
                                                 float(rand_scale), _ptr(T), _ptr(draws), _stream()), "shr_synth_pose_fwd")
    return T, draws


def mesh_render_post(T, skin_vertex_start, skin_bone, skin_wv, right_hand, camera, rand_f, faces, out_size, depth_scale,
                     noise_keys=None, sigma_xy=0.5, sigma_z=0.05, rng_state=None, src_size=640, clamp_max=100.0):
    """DepthRender + `* depth_scale` + DepthNoise as one call (shr_mesh_render_post_fwd): T [B,NB,4,4] -> [B,S,S].
    noise_keys: the samples' stream keys ([2,B] float32 view of uint32: synth_pose's draws[4:6]) or None = no noise;
    rng_state: the generator state whose call counter the launch advances (None: left alone)."""
    _check_input(T, "bone_transformations")
    _check_input(faces, "faces", torch.int32)
    if T.dim() != 4 or T.shape[2:] != (4, 4) or faces.dim() != 2 or faces.shape[1] != 3:
        raise RuntimeError("T must be [B,NB,4,4] and faces [F,3]")
    B, NB = T.shape[0], T.shape[1]
    NV = skin_vertex_start.numel() - 1
    cx, cy, fx, fy = camera
    lib = _lib.lib()
    with _on(T.device):
        depth = torch.empty((B, out_size, out_size), dtype=torch.float32, device=T.device)
        ws = dws = None
        if not lib.shr_mesh_render_one_launch(NB, NV, faces.shape[0], src_size, out_size):
            ws = torch.empty((B, NV, 4), dtype=torch.float32, device=T.device)
            dws = torch.empty_like(depth)
        _lib.check(lib.shr_mesh_render_post_fwd(_ptr(T), B, NB, NV, _ptr(skin_vertex_start), _ptr(skin_bone), _ptr(skin_wv),
                                                int(bool(right_hand)), cx, cy, fx, fy, _ptr(rand_f), _ptr(faces),
                                                faces.shape[0], src_size, out_size, clamp_max, float(depth_scale),
                                                _ptr(noise_keys), float(sigma_xy), float(sigma_z), _ptr(rng_state),
                                                _ptr(ws), _ptr(dws), _ptr(depth), _stream()), "shr_mesh_render_post_fwd")
    return depth


def heatmap_render(T, kp_start, kp_bone, kp_wv, right_hand, camera, rand_f, S, sigma, inv_k, uv_scale=1.0, d_scale=1.0):
    """Hand3DHeatmapRender in one launch (shr_heatmap_render_fwd): T [B,NB,4,4] -> (uv_hm [B,J,S,S] * uv_scale,
    d_hm [B,J,S,S] * d_scale, xyz [B,J,4]); kp_*: the key-points' CSR skin table (one entry each for the hand)."""
    _check_input(T, "bone_transformations")
    B, NB = T.shape[0], T.shape[1]
    J = kp_start.numel() - 1
    cx, cy, fx, fy = camera
    a00, a03, a11, a13 = inv_k
    with _on(T.device):
        uv = torch.empty((B, J, S, S), dtype=torch.float32, device=T.device)
        d = torch.empty((B, J, S, S), dtype=torch.float32, device=T.device)
        xyz = torch.empty((B, J, 4), dtype=torch.float32, device=T.device)
        _lib.check(_lib.lib().shr_heatmap_render_fwd(_ptr(T), B, NB, J, _ptr(kp_start), _ptr(kp_bone), _ptr(kp_wv),
                                                     int(bool(right_hand)), cx, cy, fx, fy, _ptr(rand_f), int(S), float(sigma),
                                                     float(uv_scale), float(d_scale), a00, a03, a11, a13, _ptr(uv), _ptr(d),
                                                     _ptr(xyz), _stream()), "shr_heatmap_render_fwd")
    return uv, d, xyz


def group_norm_relu_supported(x, num_groups):
    """True when the NHWC GroupNorm+ReLU kernels take this activation (CUDA fp32, channels-last)."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] > 0
            and x.is_contiguous(memory_format=torch.channels_last)
            and bool(_lib.lib().shr_group_norm_relu_supported(int(x.shape[1]), int(num_groups))))


class GroupNormReLU(torch.autograd.Function):
    """relu(group_norm(x + pre_bias, G, weight, bias, eps)) on channels-last activations
    (network/hourglass.py:28-31), one kernel per direction.  `pre_bias` [C] (optional) is the bias of the
    convolution that produced x: the convolution is then run WITHOUT its bias, and this op's backward returns
    the bias gradient with dx (two launches and one reduction less per convolution)."""

    @staticmethod
    def forward(ctx, x, weight, bias, num_groups, eps, pre_bias=None):
        N, C, H, W = x.shape
        weight, bias = weight.contiguous(), bias.contiguous()
        pre = None if pre_bias is None else pre_bias.contiguous()
        with _on(x.device):
            y = torch.empty_like(x, memory_format=torch.channels_last)
            mean = torch.empty((N, num_groups), dtype=torch.float32, device=x.device)
            rstd = torch.empty((N, num_groups), dtype=torch.float32, device=x.device)
            _lib.check(_lib.lib().shr_group_norm_relu_fwd(_ptr(x), _ptr(pre), _ptr(weight), _ptr(bias), N, C, H * W,
                                                          num_groups, float(eps), _ptr(y), _ptr(mean), _ptr(rstd),
                                                          _stream()), "shr_group_norm_relu_fwd")
        ctx.save_for_backward(x, weight, bias, mean, rstd, pre)
        ctx.num_groups = num_groups
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, mean, rstd, pre = ctx.saved_tensors
        N, C, H, W = x.shape
        dy = dy.contiguous(memory_format=torch.channels_last)
        k = 2 if pre is None else 3
        with _on(x.device):
            dx = torch.empty_like(x, memory_format=torch.channels_last)
            part = torch.empty((k, N, C), dtype=torch.float32, device=x.device)     # per-sample partials (workspace)
            dgb = torch.empty((k, C), dtype=torch.float32, device=x.device)
            _lib.check(_lib.lib().shr_group_norm_relu_bwd(_ptr(x), _ptr(pre), _ptr(dy), _ptr(weight), _ptr(bias),
                                                          _ptr(mean), _ptr(rstd), N, C, H * W, ctx.num_groups, _ptr(dx),
                                                          _ptr(part[0]), _ptr(part[1]),
                                                          _ptr(part[2]) if pre is not None else None,
                                                          _ptr(dgb[0]), _ptr(dgb[1]),
                                                          _ptr(dgb[2]) if pre is not None else None,
                                                          _stream()), "shr_group_norm_relu_bwd")
        return dx, dgb[0], dgb[1], None, None, (dgb[2] if pre is not None else None)


FUSED_GROUP_NORM_RELU = True    # False: always torch's group_norm + relu (A/B measurements)


def group_norm_relu(x, gn, pre_bias=None):
    """F.relu(gn(x + pre_bias)) for an nn.GroupNorm `gn`: the NHWC kernels when they apply, torch otherwise."""
    if FUSED_GROUP_NORM_RELU and gn.affine and group_norm_relu_supported(x, gn.num_groups):
        return GroupNormReLU.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, pre_bias)
    if pre_bias is not None:
        x = x + pre_bias.view(1, -1, 1, 1)
    return torch.nn.functional.relu(gn(x))


def conv_then_group_norm_relu(x, conv, gn):
    """relu(gn(conv(x))) with the convolution's bias folded into the normalisation kernel when that kernel will
    take conv's output (NHWC fp32 on the GPU, supported channel counts): conv runs bias-free."""
    # (F.conv2d below bypasses conv.forward: only for a plain zero-padded convolution without hooks or
    # parametrizations; the kernel reads the bias 16 bytes at a time)
    plain = (conv.padding_mode == 'zeros' and not conv._forward_hooks and not conv._forward_pre_hooks
             and not getattr(conv, 'parametrizations', None) and conv.bias is not None
             and conv.bias.data_ptr() % 16 == 0 and conv.bias.is_contiguous())
    if (FUSED_GROUP_NORM_RELU and gn.affine and plain and x.is_cuda and x.dtype == torch.float32
            and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and bool(_lib.lib().shr_group_norm_relu_supported(int(conv.out_channels), int(gn.num_groups)))):
        y = torch.nn.functional.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
        if group_norm_relu_supported(y, gn.num_groups):
            return GroupNormReLU.apply(y, gn.weight, gn.bias, gn.num_groups, gn.eps, conv.bias)
        return torch.nn.functional.relu(gn(y + conv.bias.view(1, -1, 1, 1)))
    return group_norm_relu(conv(x), gn)


def heatmap_paint(uvd, S, sigma, inv_k, uv_scale=1.0, d_scale=1.0):
    """uvd [B,J,4] heat-map-space key-points -> (uv_hm [B,J,S,S] * uv_scale, d_hm [B,J,S,S] * d_scale,
    xyz [B,J,4] = inv_k @ uvd): HeatmapRender + InverseOthographicalProjection in one launch."""
    _check_input(uvd, "uvd_points")
    if uvd.dim() != 3 or uvd.shape[2] != 4:
        raise RuntimeError("uvd_points must be [B,J,4]")
    B, J = uvd.shape[0], uvd.shape[1]
    a00, a03, a11, a13 = (float(inv_k[0][0]), float(inv_k[0][3]), float(inv_k[1][1]), float(inv_k[1][3]))
    with _on(uvd.device):
        uv = torch.empty((B, J, S, S), dtype=torch.float32, device=uvd.device)
        d = torch.empty((B, J, S, S), dtype=torch.float32, device=uvd.device)
        xyz = torch.empty((B, J, 4), dtype=torch.float32, device=uvd.device)
        _lib.check(_lib.lib().shr_heatmap_paint(_ptr(uvd), B * J, int(S), float(sigma), float(uv_scale), float(d_scale),
                                                a00, a03, a11, a13, _ptr(uv), _ptr(d), _ptr(xyz), _stream()),
                   "shr_heatmap_paint")
    return uv, d, xyz


def depth_noise(depth, sigma_xy, sigma_z, generator=None):
    """DepthNoise on [B,H,W] scaled depth: one torch.randn([3,B,H,W]) + one launch."""
    _check_input(depth, "depth")
    B, H, W = depth.shape
    with _on(depth.device):
        normal3 = torch.randn((3, B, H, W), dtype=torch.float32, device=depth.device, generator=generator)
        out = torch.empty_like(depth)
        _lib.check(_lib.lib().shr_depth_noise(_ptr(depth), _ptr(normal3), B, H, W, float(sigma_xy), float(sigma_z),
                                              _ptr(out), _stream()), "shr_depth_noise")
    return out


def soft_argmax_supported(hm, J):
    """True when the soft-argmax kernels take this raw network output (CUDA fp32 [N,2J,h,w], pixels
    linear in memory: NCHW or channels-last, also as a batch slice)."""
    if not (hm.is_cuda and hm.dtype == torch.float32 and hm.dim() == 4 and hm.shape[1] == 2 * J and hm.shape[0] > 0):
        return False
    h, w = hm.shape[2], hm.shape[3]
    if hm.stride(2) != w * hm.stride(3) or hm.stride(3) < 1 or hm.stride(1) < 1:
        return False
    return bool(_lib.lib().shr_soft_argmax_supported(int(J), int(h), int(w)))


class SoftArgmaxXYZ(torch.autograd.Function):
    """hm [N,2J,h,w] (uv heat-maps | depth heat-maps) -> xyz [N,J,3]: RecoverXYZCoordinateFromHeatmap
    (network/util_modules.py:164-201) in one launch per direction."""

    @staticmethod
    def forward(ctx, hm, J, cx, cy, fx, fy, depth_scale_inv):
        N, _, h, w = hm.shape
        with _on(hm.device):
            xyz = torch.empty((N, J, 3), dtype=torch.float32, device=hm.device)
            _lib.check(_lib.lib().shr_soft_argmax_fwd(_ptr(hm), hm.stride(0), hm.stride(1), hm.stride(3), N, J, h, w,
                                                      float(cx), float(cy), float(fx), float(fy), float(depth_scale_inv),
                                                      _ptr(xyz), _stream()), "shr_soft_argmax_fwd")
        ctx.save_for_backward(hm)
        ctx.args = (J, cx, cy, fx, fy, depth_scale_inv)
        return xyz

    @staticmethod
    def backward(ctx, grad_xyz):
        (hm,) = ctx.saved_tensors
        J, cx, cy, fx, fy, ds = ctx.args
        N, _, h, w = hm.shape
        grad_xyz = grad_xyz.contiguous()
        with _on(hm.device):
            grad_hm = torch.empty_strided(hm.shape, hm.stride(), dtype=torch.float32, device=hm.device)
            _lib.check(_lib.lib().shr_soft_argmax_bwd(_ptr(hm), hm.stride(0), hm.stride(1), hm.stride(3), N, J, h, w,
                                                      float(cx), float(cy), float(fx), float(fy), float(ds),
                                                      _ptr(grad_xyz), _ptr(grad_hm), _stream()), "shr_soft_argmax_bwd")
        return grad_hm, None, None, None, None, None, None


class PairLosses(torch.autograd.Function):
    """(joints [B,P,3] with P >= J: the first J points of a sample are its sphere centres, as the reference's
    `joints.view(B, -1, 3)` indexing implies) -> (collision loss, bone-length loss) of mesh/render.py:145-206
    with the reference's reductions (sum; mean of the two hinges over [B,K]).  One launch; the backward only
    scales the unit gradients the kernel already produced."""

    @staticmethod
    def forward(ctx, joints, J, num_palm, per_finger, min_dist_sq, bone_a, bone_b, bone_min_sq, bone_max_sq):
        joints = joints.contiguous()
        B, P = joints.shape[0], joints.shape[1]
        K = bone_a.numel()
        with _on(joints.device):
            sums = torch.empty((3, B), dtype=torch.float32, device=joints.device)
            grads = torch.empty((3, B, J, 3), dtype=torch.float32, device=joints.device)
            _lib.check(_lib.lib().shr_pair_losses(_ptr(joints), P * 3, B, J, int(num_palm), int(per_finger),
                                                  float(min_dist_sq), _ptr(bone_a), _ptr(bone_b), _ptr(bone_min_sq),
                                                  _ptr(bone_max_sq), K, _ptr(sums[0]), _ptr(sums[1]), _ptr(sums[2]),
                                                  _ptr(grads[0]), _ptr(grads[1]), _ptr(grads[2]), _stream()),
                       "shr_pair_losses")
        ctx.save_for_backward(grads)
        ctx.meta = (B, P, J, K)
        tot = sums.sum(dim=1)
        return tot[0], (tot[1] + tot[2]) / float(B * K)

    @staticmethod
    def backward(ctx, g_coll, g_bone):
        (grads,) = ctx.saved_tensors
        B, P, J, K = ctx.meta
        g = grads[0] * g_coll + (grads[1] + grads[2]) * (g_bone / float(B * K))
        if P > J:
            full = torch.zeros((B, P, 3), dtype=torch.float32, device=g.device)
            full[:, :J] = g
            g = full
        return g, None, None, None, None, None, None, None, None


class MultiviewConsistency(torch.autograd.Function):
    """(camera_poses [B,V,4,4], joints [B,V,J,3]) -> MSELoss(median over the views, canonical points): the
    reference's MultiviewConsistencyLoss without heat-map weights (mesh/multiview_utility.py:138-167), one launch;
    the backward only scales the unit gradient the kernel already produced."""

    @staticmethod
    def forward(ctx, camera_poses, joints):
        cam = camera_poses.detach().contiguous().float()
        joints = joints.contiguous().float()
        _check_input(cam, "camera_poses")
        _check_input(joints, "joints")
        B, V, J = joints.shape[0], joints.shape[1], joints.shape[2]
        if cam.shape != (B, V, 4, 4) or joints.dim() != 4 or joints.shape[3] != 3:
            raise RuntimeError("expected camera_poses [B,V,4,4] and joints [B,V,J,3]")
        want = ctx.needs_input_grad[1]
        with _on(joints.device):
            sums = torch.empty(B, dtype=torch.float32, device=joints.device)
            grad = torch.empty((B, V, J, 3), dtype=torch.float32, device=joints.device) if want else None
            _lib.check(_lib.lib().shr_mv_consistency(_ptr(cam), _ptr(joints), B, V, J, _ptr(sums), _ptr(grad), _stream()),
                       "shr_mv_consistency")
        if want:
            ctx.save_for_backward(grad)
        ctx.count = float(B * V * J * 3)
        return sums.sum() / ctx.count

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return None, grad * (g / ctx.count)


def mv_consistency_supported(camera_poses, joints):
    return (joints.is_cuda and camera_poses.is_cuda and joints.dim() == 4 and joints.shape[-1] == 3
            and joints.shape[1] <= 8 and joints.shape[2] <= 64 and joints.dtype == torch.float32)


def depth_resample(depth, sample_ratio, kernel_size, generator=None):
    """DepthResample on [N,H,W] scaled depth: one torch.rand + one launch -> [N,1,H,W]."""
    _check_input(depth, "depth")
    N, H, W = depth.shape
    with _on(depth.device):
        uniform = torch.rand((N, H, W), dtype=torch.float32, device=depth.device, generator=generator)
        out = torch.empty((N, 1, H, W), dtype=torch.float32, device=depth.device)
        _lib.check(_lib.lib().shr_depth_resample(_ptr(depth), _ptr(uniform), N, H, W, float(sample_ratio),
                                                 int(kernel_size), _ptr(out), _stream()), "shr_depth_resample")
    return out
