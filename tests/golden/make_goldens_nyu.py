#!/usr/bin/env python3
"""Golden vectors for the NYU crop generator, by RUNNING the imported reference on CPU over a synthetic
NYU-layout directory (tests/nyu_synth.py writes it from a seed; the test regenerates the same frames).

    python tests/golden/make_goldens_nyu.py  ->  tests/golden/g10_nyu_generator.npz

Reference entry points (file:line in /root/reference):
  dataset/nyu_generator.py:15-130   NyuDatasetGenerator (load, crop, camera poses, shard files)
  dataset/utils.py:70-145           crop_dm, estimate_rigid_transformation
Accommodations: the module parses sys.argv at import (nyu_generator.py:132-134) -> argv is emptied for the
import; it imports `utils` from its own directory -> dataset/ is put on sys.path.  The reference is fixed at
64 x 64 (its img_size attribute is set after construction for the 128 case, which its own code then uses
everywhere: crop_dm and the shard directory name are computed from it at call time except npy_dir)."""
import os
import pickle
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from _refimport import import_reference, REF  # noqa: E402
from nyu_synth import write_synthetic_nyu     # noqa: E402


def main():
    import_reference()
    sys.path.insert(0, os.path.join(REF, "dataset"))
    argv, sys.argv = sys.argv, [sys.argv[0]]
    try:
        import nyu_generator as ref
    finally:
        sys.argv = argv
    out = {}
    with tempfile.TemporaryDirectory() as root:
        write_synthetic_nyu(root, "train", frames=3, seed=0)
        gen = ref.NyuDatasetGenerator(root, "train")
        gen.create_npy_dataset_from_indices("mv_data_0", [0, 1, 2])
        npy = os.path.join(root, "npy-64", "train")
        shape = pickle.load(open(os.path.join(npy, "mv_data_0_shape.pkl"), "rb"))
        out["dms64"] = np.array(np.memmap(os.path.join(npy, "mv_data_0_dms.bat"), dtype="float32", mode="r",
                                          shape=tuple(shape["dms"])))
        out["joint_poses"] = np.load(os.path.join(npy, "mv_data_0_joint_poses.npy"))
        out["camera_poses"] = np.load(os.path.join(npy, "mv_data_0_camera_poses.npy"))
        # the same frames cropped at 128 x 128 by the reference's own functions
        gen.img_size = (128, 128)
        dms, ann = gen.load_sample_from_file(1)
        c128, _ = gen.crop_sample(dms, ann)
        out["dms128_frame1"] = c128.astype(np.float32)
    print({k: (v.shape, v.dtype) for k, v in out.items()}, "foreground px @64:", int((out["dms64"] < 100).sum()))
    np.savez_compressed(os.path.join(HERE, "g10_nyu_generator.npz"), **out)


if __name__ == "__main__":
    main()
