"""Procedural sampler of the 26-D hand pose vector (benchmark / synthetic-branch
inputs).

Mirrors the reference's ``JointAngleDataset`` (dataset/joint_angle.py:7-236):
same distribution, same ``__getitem__``/``__len__`` surface, and the same
sequence of ``torch.rand(1)`` draws, so that under the same ``torch.manual_seed``
it yields the same vectors (checked against tests/golden/g3_batch256.npz).

Pose layout ([0:6] palm Euler xyz + translation; four values per finger =
abduct, flex1, flex2, flex3 at 6 index, 10 middle, 14 ring, 18 pinky, 22 thumb)
is the one HandTransformationMat consumes (mesh/kinematicsTransformation.py:169-175).
"""
from math import pi

import torch
import torch.utils.data as data

NUM_PARAMETER = 26
FINGER_BASE = {"index": 6, "middle": 10, "ring": 14, "pinky": 18, "thumb": 22}
_REST_CURL = (-0.2, -0.4, -0.34)          # flex offsets of a curled finger (:48,:69,:91)


def _u():
    return torch.rand(1)


def _deg(lo, span):
    """U(lo, lo+span) degrees -> radians, evaluated as ((r*span + lo) * pi) / 180."""
    return (_u() * span + lo) * pi / 180


def _jitter():
    return (_u() * 20 - 10) * pi / 180        # +-10 degrees (:43-44)


def _affine3(scale, shift):
    return torch.tensor([_u() * scale[0] - shift[0], _u() * scale[1] - shift[1],
                         _u() * scale[2] - shift[2]]).type(torch.float)


def _straight():                               # :112-116
    return _affine3((0.25, 0.4, 0.34), (0.25, 0.4, 0.34))


def _open():                                   # :106-110
    return _affine3((0.25, 0.4, 0.34), (0.1, 0.1, 0.1))


def _curl(first, rest):
    """Three coupled curls (:42-104): joint k gets its own curl plus a share of
    its neighbours'.  `first`/`rest` draw the base angle of curl 1 / curls 2,3."""
    f1, f2, f3 = _REST_CURL
    c = first() + _jitter()
    f1 = f1 + 1.0 * c
    f2 = f2 + 0.2 * c
    c = rest() + _jitter()
    f1 = f1 + 0.2 * c
    f2 = f2 + 1.0 * c
    f3 = f3 + 0.7 * c
    c = rest() + _jitter()
    f2 = f2 + 0.2 * c
    f3 = f3 + 1.0 * c
    return torch.tensor([f1, f2, f3]).type(torch.float)


def _half_open():                              # :85-104
    return _curl(lambda: (_u() * 30) * pi / 180, lambda: _deg(60, 30))


def _pinching():                               # :63-82
    return _curl(lambda: _deg(60, 30), lambda: _deg(5, 30))


def _closed():                                 # :42-61
    return _curl(lambda: _deg(60, 30), lambda: _deg(60, 30))


_OPENISH = (_straight, _open, _half_open)
_CLOSEDISH = (_pinching, _closed)
_ANY = _OPENISH + _CLOSEDISH


def _pick(table):
    return table[int(_u() * len(table))]()


# hand-level flex pattern -> generator table per finger (index, middle, ring, pinky) (:160-214)
_O, _C = _OPENISH, _CLOSEDISH
_PATTERNS = {5: (_O, _C, _C, _C), 6: (_C, _C, _C, _O), 7: (_O, _O, _C, _C), 8: (_C, _O, _O, _O),
             9: (_ANY, _ANY, _ANY, _ANY)}


def sample_pose():
    p = torch.zeros(NUM_PARAMETER)
    # palm (:22-29): rotation about x,z in (-3.14,3.14), about y in (-3.14,0); translation mm
    p[0:6] = torch.tensor([_u() * 6.28 - 3.14, -_u() * 3.14, _u() * 6.28 - 3.14,
                           _u() * 30 - 15, _u() * 30 - 15, _u() * 50 - 35])
    # finger spread (:32-40)
    spread = (_u() - 0.35) / 1.55
    wiggle = lambda: (_u() * 10 - 5) * pi / 180   # noqa: E731
    abduct = [1.55 * (spread + wiggle()), 0.75 * (spread + wiggle()),
              -0.75 * (spread + wiggle()), -2.2 * (spread + wiggle())]
    # thumb (:118-129)
    flex = _u() * 0.35 - 0.25 if _u() < 0.5 else _u() * 0.6 + 0.1
    flex3 = _u() * 2 - 1.7
    p[22:26] = torch.tensor([_u() - 0.5, flex, 0.25 * flex, flex3]).type(torch.float)
    for k, name in enumerate(("index", "middle", "ring", "pinky")):
        p[FINGER_BASE[name]] = abduct[k]
    mode = int(_u() * 10)
    if mode < 5:
        flexes = [_ANY[mode]() for _ in range(4)]
    else:
        flexes = [_pick(t) for t in _PATTERNS[mode]]
    for k, name in enumerate(("index", "middle", "ring", "pinky")):
        b = FINGER_BASE[name]
        p[b + 1:b + 4] = flexes[k]
    return p


class JointAngleDataset(data.Dataset):
    def __init__(self):
        super().__init__()
        self.num_parameter = NUM_PARAMETER

    def __getitem__(self, index):
        return sample_pose()

    def __len__(self):
        return 400000


def sample_poses(n, seed=None):
    """[n,26] poses; with `seed`, reseeds torch's CPU generator first (the
    benchmark uses seed 0: SURVEY section 8d)."""
    if seed is not None:
        torch.manual_seed(seed)
    return torch.stack([sample_pose() for _ in range(n)])


# --------------------------------------------------------------------------- vectorised sampler
_DEG = pi / 180
# generator kinds per (sample, finger): 0 straight, 1 open, 2 half-open, 3 pinching, 4 closed (= _ANY's order)
_TABLE_KINDS = {"O": (0, 1, 2), "C": (3, 4), "A": (0, 1, 2, 3, 4)}
_PATTERN_TABLES = {5: "OCCC", 6: "CCCO", 7: "OOCC", 8: "COOO", 9: "AAAA"}


def sample_poses_batched(n, generator=None):
    """[n,26] poses from the SAME distribution as n calls of sample_pose() (dataset/joint_angle.py:7-236), drawn
    with a handful of [n, .] tensor operations instead of ~55 torch.rand(1) calls per pose (48 poses: 7 ms of host
    time per training step against 0.15 ms).  The draw SEQUENCE differs from sample_pose's -- as it does between the
    reference's own DataLoader workers (network/engine.py:328-329: shuffle=True, num_workers=1); the pinned sequence
    (tests/golden/g3_batch256.npz) stays with sample_pose / sample_poses.  tests/test_network_cpu.py compares the
    two samplers' moments, ranges and per-finger generator frequencies."""
    def u(*shape):
        return torch.rand(*shape, generator=generator)
    p = torch.zeros(n, NUM_PARAMETER)
    # palm (:22-29)
    p[:, 0] = u(n) * 6.28 - 3.14
    p[:, 1] = -u(n) * 3.14
    p[:, 2] = u(n) * 6.28 - 3.14
    p[:, 3] = u(n) * 30 - 15
    p[:, 4] = u(n) * 30 - 15
    p[:, 5] = u(n) * 50 - 35
    # finger spread (:32-40)
    spread = ((u(n) - 0.35) / 1.55).unsqueeze(1)
    wiggle = (u(n, 4) * 10 - 5) * _DEG
    p[:, [6, 10, 14, 18]] = torch.tensor([1.55, 0.75, -0.75, -2.2]) * (spread + wiggle)
    # thumb (:118-129)
    flex = torch.where(u(n) < 0.5, u(n) * 0.35 - 0.25, u(n) * 0.6 + 0.1)
    p[:, 22] = u(n) - 0.5
    p[:, 23] = flex
    p[:, 24] = 0.25 * flex
    p[:, 25] = u(n) * 2 - 1.7
    # which generator each of the four fingers uses (:160-214): modes 0-4 = one generator for all four,
    # modes 5-9 = a uniform pick from the pattern's table per finger
    mode = (u(n) * 10).long().clamp_(max=9)
    pick = u(n, 4)
    kind = mode.unsqueeze(1).expand(n, 4).clone()
    for m, tables in _PATTERN_TABLES.items():
        rows = mode == m
        if bool(rows.any()):
            for k, t in enumerate(tables):
                opts = torch.tensor(_TABLE_KINDS[t])
                kind[rows, k] = opts[(pick[rows, k] * len(opts)).long().clamp_(max=len(opts) - 1)]
    # the five generators, evaluated for every (sample, finger) and selected
    scale = torch.tensor([0.25, 0.4, 0.34])
    straight = u(n, 4, 3) * scale - scale                                    # :112-116
    opened = u(n, 4, 3) * scale - 0.1                                        # :106-110

    def curl(first_lo, first_span, rest_lo, rest_span):                     # :42-104
        jit = (u(n, 4, 3) * 20 - 10) * _DEG
        c1 = (u(n, 4) * first_span + first_lo) * _DEG + jit[..., 0]
        c2 = (u(n, 4) * rest_span + rest_lo) * _DEG + jit[..., 1]
        c3 = (u(n, 4) * rest_span + rest_lo) * _DEG + jit[..., 2]
        f1, f2, f3 = _REST_CURL
        return torch.stack([f1 + c1 + 0.2 * c2, f2 + 0.2 * c1 + c2 + 0.2 * c3, f3 + 0.7 * c2 + c3], -1)
    variants = torch.stack([straight, opened, curl(0, 30, 60, 30), curl(60, 30, 5, 30), curl(60, 30, 60, 30)], 2)
    flexes = torch.gather(variants, 2, kind.view(n, 4, 1, 1).expand(n, 4, 1, 3)).squeeze(2)      # [n,4,3]
    for k, name in enumerate(("index", "middle", "ring", "pinky")):
        b = FINGER_BASE[name]
        p[:, b + 1:b + 4] = flexes[:, k]
    return p
