"""List or convert a TensorFlow checkpoint of the reference (tf.train.Checkpoint files: `checkpoint`, ckpt-N.index,
ckpt-N.data-00000-of-00001) without TensorFlow.

    python tools/convert_checkpoint.py checkpoints/hyper/a6b3/                 # list variables, shapes, dtypes
    python tools/convert_checkpoint.py checkpoints/hyper/a6b3/ out_dir/        # write out_dir/weights.npz (model variables only)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                                 # noqa: E402

from pcgcv1_amd import checkpoint, tf_bundle                       # noqa: E402


def main(argv):
    if not argv:
        raise SystemExit(__doc__)
    src = argv[0]
    prefix = tf_bundle.latest_checkpoint(src) if os.path.isdir(src) else src
    if prefix is None or not os.path.exists(prefix + ".index"):
        raise SystemExit("no checkpoint found at %r" % src)
    raw = tf_bundle.read_bundle(prefix)
    print("%s: %d tensors" % (prefix, len(raw)))
    for k in sorted(raw):
        print("  %-90s %-18s %s" % (k, raw[k].shape, raw[k].dtype))
    if len(argv) > 1:
        w = checkpoint._from_bundle(prefix)
        checkpoint.save(w, argv[1])
        print("wrote %s (%d model variables, %d parameters)" % (os.path.join(argv[1], "weights.npz"), len(w),
                                                              sum(int(np.prod(v.shape)) for v in w.values())))


if __name__ == "__main__":
    main(sys.argv[1:])
