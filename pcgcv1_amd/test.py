"""Command line of the reference's test.py (24-45 flags, 61-115 dispatch), same flags and defaults:

    python -m pcgcv1_amd.test compress  X.ply  --ckpt_dir=checkpoints/hyper/a6b3/
    python -m pcgcv1_amd.test decompress compressed/X --ckpt_dir=checkpoints/hyper/a6b3/

compress writes ./compressed/<basename>.{strings,strings_head,strings_hyper,pointnums,cubepos};
decompress writes <name>_rec.ply.  --ckpt_dir additionally accepts "synthetic[:seed[:profile]]"
(checkpoint.py).  --gpu=0 is rejected: this build has no CPU path.
"""
import argparse
import importlib
import os


def parse_args(argv=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("command", choices=["compress", "decompress"],
                        help="'compress' reads a point cloud (.ply) and writes compressed binary files; "
                             "'decompress' reads binary files and reconstructs the point cloud (.ply).")
    parser.add_argument("input", nargs="?", help="Input filename.")
    parser.add_argument("output", nargs="?", help="Output filename.")
    parser.add_argument("--mode", type=str, default='hyper', dest="mode", help='factorized entropy model or hyper prior')
    parser.add_argument("--modelname", default="models.model_voxception", dest="modelname",
                        help="(model_simple, model_voxception)")
    parser.add_argument("--ckpt_dir", type=str, default='', dest="ckpt_dir", help='checkpoint')
    parser.add_argument("--scale", type=float, default=1.0, dest="scale", help="scaling factor.")
    parser.add_argument("--cube_size", type=int, default=64, dest="cube_size", help="size of partitioned cubes.")
    parser.add_argument("--min_num", type=int, default=64, dest="min_num", help="minimum number of points in a cube.")
    parser.add_argument("--rho", type=float, default=1.0, dest="rho",
                        help="ratio of the numbers of output points to the number of input points.")
    parser.add_argument("--gpu", type=int, default=1, dest="gpu", help="use gpu (1) or not (0).")
    args = parser.parse_args(argv)
    print(args)
    return args


def _import_model(name):
    if name.startswith("models."):
        name = "pcgcv1_amd." + name
    return importlib.import_module(name)


def main(argv=None):
    args = parse_args(argv)
    if args.gpu != 1:
        raise SystemExit("--gpu=0: this build runs the hot path on an MI355X only (no CPU fallback)")
    from .process import preprocess, postprocess
    from .transform import compress_hyper, decompress_hyper, compress_factorized, decompress_factorized
    from .dataprocess import inout_bitstream as bs
    model = _import_model(args.modelname)
    if args.command == "compress":
        if not args.output:
            args.output = os.path.split(args.input)[-1][:-4]
        cubes, cube_positions, points_numbers = preprocess(args.input, args.scale, args.cube_size, args.min_num)
        if args.mode == "factorized":
            strings, min_v, max_v, shape = compress_factorized(cubes, model, args.ckpt_dir, verbose=True)
            bs.write_binary_files_factorized(args.output, strings, points_numbers, cube_positions, min_v, max_v, shape,
                                             rootdir='./compressed')
        else:
            (y_strings, y_min_vs, y_max_vs, y_shape, z_strings, z_min_v, z_max_v, z_shape) = compress_hyper(
                cubes, model, args.ckpt_dir, verbose=True)
            bs.write_binary_files_hyper(args.output, y_strings, z_strings, points_numbers, cube_positions, y_min_vs,
                                        y_max_vs, y_shape, z_min_v, z_max_v, z_shape, rootdir='./compressed')
    else:
        rootdir, filename = os.path.split(args.input)
        if not args.output:
            args.output = filename + "_rec.ply"
        if args.mode == "factorized":
            strings, points_numbers, cube_positions, min_v, max_v, shape = bs.read_binary_files_factorized(filename, rootdir)
            cubes = decompress_factorized(strings, min_v, max_v, shape, model, args.ckpt_dir, verbose=True)
        else:
            (y_strings, z_strings, points_numbers, cube_positions, y_min_vs, y_max_vs, y_shape, z_min_v, z_max_v,
             z_shape) = bs.read_binary_files_hyper(filename, rootdir)
            cubes = decompress_hyper(y_strings, y_min_vs, y_max_vs, y_shape, z_strings, z_min_v, z_max_v, z_shape, model,
                                     args.ckpt_dir, verbose=True)
        postprocess(args.output, cubes, points_numbers, cube_positions, args.scale, args.cube_size, args.rho)


if __name__ == "__main__":
    main()
