"""Command line of the reference's test.py (24-45 flags, 61-115 dispatch), same flags and defaults:

    python -m pcgcv1_amd.test compress  X.ply  --ckpt_dir=checkpoints/hyper/a6.00b3.00
    python -m pcgcv1_amd.test decompress compressed/X --ckpt_dir=checkpoints/hyper/a6.00b3.00

compress writes ./compressed/<basename>.{strings,strings_head,strings_hyper,pointnums,cubepos};
decompress writes <name>_rec.ply.  --ckpt_dir additionally accepts "synthetic[:seed[:profile]]"
(checkpoint.py).  --gpu=0 is rejected: this build has no CPU path; --gpu=N shards the cubes over N GPUs
(one rank per GPU over RCCL, pcgcv1_amd/sharding.py; same files as one GPU).
"""
import argparse
import importlib
import os
import time


# flag, type, default, meaning — names and defaults are the reference's (test.py:24-45)
_FLAGS = [
    ("mode", str, "hyper", "entropy model: 'hyper' (hyperprior) or 'factorized'"),
    ("modelname", str, "models.model_voxception", "module with the transforms: models.model_voxception | models.model_simple"),
    ("ckpt_dir", str, "", "TensorFlow checkpoint directory, weights.npz directory, or synthetic[:seed[:profile]]"),
    ("scale", float, 1.0, "coordinates are multiplied by this before partitioning (and divided back on output)"),
    ("cube_size", int, 64, "edge of the cubes the cloud is cut into"),
    ("min_num", int, 64, "cubes with fewer points are dropped"),
    ("rho", float, 1.0, "output points per cube = rho x the stored point count"),
    ("gpu", int, 1, "GPUs to use: 1 = this process; N > 1 = the cube list sharded over N ranks, one per GPU (started here "
                    "unless a launcher already set WORLD_SIZE); 0 is refused: there is no CPU path"),
]


def parse_args(argv=None):
    ap = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument("command", choices=("compress", "decompress"), help="compress: .ply -> ./compressed/<name>.*; "
                                                                         "decompress: those files -> <name>_rec.ply")
    ap.add_argument("input", nargs="?", help="point cloud (.ply) or compressed file stem")
    ap.add_argument("output", nargs="?", help="output stem / .ply (derived from the input when omitted)")
    for name, typ, default, meaning in _FLAGS:
        ap.add_argument("--" + name, type=typ, default=default, help=meaning)
    args = ap.parse_args(argv)
    print(args)
    return args


def _import_model(name):
    if name.startswith("models."):
        name = "pcgcv1_amd." + name
    return importlib.import_module(name)


def _report(model, ckpt_dir, t0, also=""):
    import torch
    from .transform import get_codec
    torch.cuda.synchronize()
    p = get_codec(model, ckpt_dir).last_path
    print("{}{}: {}s ({} cubes, {} host pipeline{})".format(p.get("call"), also, round(time.time() - t0, 4), p.get("cubes"),
                                                           p.get("pipelines"), "" if p.get("pipelines") == 1 else "s"))


def _main_sharded(args, world):
    """One process per GPU (`python -m torch.distributed.run --nproc-per-node N -m pcgcv1_amd.test ...`): the cube
    list is split over the ranks (pcgcv1_amd/sharding.py), rank 0 reads and writes the files.  Same files as one GPU."""
    import torch
    import torch.distributed as dist
    from . import sharding
    from .dataprocess import inout_bitstream as bs
    from .process import postprocess_masks, preprocess
    if args.mode != "hyper":
        raise SystemExit("multi-GPU runs are implemented for --mode=hyper")
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    if not dist.is_initialized():
        dist.init_process_group(os.environ.get("PCGC_BACKEND", "nccl"))
    ops = sharding.HipOps(_import_model(args.modelname), args.ckpt_dir)
    if args.command == "compress":
        if not args.output:
            args.output = os.path.split(args.input)[-1][:-4]
        # every rank parses + partitions the cloud (host, cheap) but voxelises and uploads only its own block of cubes
        cubes, cube_positions, nums_local = preprocess(args.input, args.scale, args.cube_size, args.min_num, verbose=rank == 0,
                                                       block=(rank, world))
        stream = sharding.compress_hyper_sharded(cubes, ops, total=len(cube_positions), points_numbers=nums_local)
        if rank == 0:
            y_strings, y_min_vs, y_max_vs, y_shape, z_strings, z_min_v, z_max_v, z_shape, points_numbers = stream
            bs.write_binary_files_hyper(args.output, y_strings, z_strings, points_numbers, cube_positions, y_min_vs, y_max_vs,
                                        y_shape, z_min_v, z_max_v, z_shape, rootdir='./compressed')
    else:
        rootdir, filename = os.path.split(args.input)
        if not args.output:
            args.output = filename + "_rec.ply"
        stream = nums = pos = None
        if rank == 0:
            (y_strings, z_strings, nums, pos, y_min_vs, y_max_vs, y_shape, z_min_v, z_max_v,
             z_shape) = bs.read_binary_files_hyper(filename, rootdir)
            stream = (y_strings, y_min_vs, y_max_vs, y_shape, z_strings, z_min_v, z_max_v, z_shape)
        masks = sharding.decompress_hyper_sharded(stream, ops, points_numbers=nums, rho=args.rho)
        if rank == 0:
            postprocess_masks(args.output, masks, pos, args.scale, args.cube_size)
    dist.barrier()


def _self_launch(argv, n):
    """--gpu=N without a launcher: the N ranks as a child `torch.distributed.run` of this module, started before this
    process has touched the GPU; exits with the child's code (non-zero if any rank failed)."""
    import subprocess
    import sys
    # --standalone: torchrun hosts the c10d rendezvous itself on a port IT binds and keeps (a port probed here by bind / close
    # could be taken by another process before the ranks meet)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(n),
           "-m", "pcgcv1_amd.test"] + list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


def main(argv=None):
    args = parse_args(argv)
    if args.gpu < 1:
        raise SystemExit("--gpu=0: this build runs the hot path on an MI355X only (no CPU fallback)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpu > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_self_launch(argv, args.gpu))
    if args.gpu > 1 and world != args.gpu:                 # e.g. --gpu=8 under `torchrun --nproc-per-node 1`: never silently one GPU
        raise SystemExit("WORLD_SIZE=%d but --gpu=%d" % (world, args.gpu))
    if world > 1:
        return _main_sharded(args, world)
    from .process import preprocess, postprocess, StreamedPostprocess
    from .transform import compress_hyper, decompress_hyper, compress_factorized, decompress_factorized
    from .dataprocess import inout_bitstream as bs
    model = _import_model(args.modelname)
    stage_times = os.environ.get("PCGC_STAGE_TIMES", "0") == "1"
    if args.command == "compress":
        if not args.output:
            args.output = os.path.split(args.input)[-1][:-4]
        cubes, cube_positions, points_numbers = preprocess(args.input, args.scale, args.cube_size, args.min_num)
        if args.mode == "factorized":
            strings, min_v, max_v, shape = compress_factorized(cubes, model, args.ckpt_dir, verbose=True)
            bs.write_binary_files_factorized(args.output, strings, points_numbers, cube_positions, min_v, max_v, shape,
                                             rootdir='./compressed')
        else:
            # the batched, two-pipeline path bench.py times; PCGC_STAGE_TIMES=1 prints the reference's per-stage times
            # instead (transform.py:121-171), which serialises the stages
            t0 = time.time()
            from . import _lib
            # host work under the GPU's: the cube positions are coded (0.6 ms of Python, interpreter lock held) once the encoder's
            # pipeline threads have queued their kernels and sit in event waits — started at once it delayed THEIR start by as much
            # (tools/exp/t_cli_timeline.py)
            def _cubepos():
                time.sleep(0.003)
                return bs.encode_cube_positions(cube_positions)
            cubepos = _lib.workers("job").submit(_cubepos)
            (y_strings, y_min_vs, y_max_vs, y_shape, z_strings, z_min_v, z_max_v, z_shape) = compress_hyper(
                cubes, model, args.ckpt_dir, verbose=stage_times)
            _report(model, args.ckpt_dir, t0)
            bs.write_binary_files_hyper(args.output, y_strings, z_strings, points_numbers, cube_positions, y_min_vs,
                                        y_max_vs, y_shape, z_min_v, z_max_v, z_shape, rootdir='./compressed',
                                        cubepos=cubepos.result())
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
            t0 = time.time()
            # the tail (top-k, points, text, file) follows the decoder slice by slice instead of waiting for the last cube;
            # PCGC_STREAM_TAIL=0 / --scale != 1 / PCGC_STAGE_TIMES=1: postprocess on the whole batch, as the reference does
            tail = None
            if args.scale == 1 and not stage_times and os.environ.get("PCGC_STREAM_TAIL", "1") != "0":
                tail = StreamedPostprocess(args.output, points_numbers, cube_positions, args.scale, args.cube_size, args.rho)
            cubes = decompress_hyper(y_strings, y_min_vs, y_max_vs, y_shape, z_strings, z_min_v, z_max_v, z_shape, model,
                                     args.ckpt_dir, verbose=stage_times, on_slice=tail)
            if tail is not None:
                print('===== Post process =====')
                tail.finish()
                _report(model, args.ckpt_dir, t0, " + post process")
                return
            _report(model, args.ckpt_dir, t0)
        postprocess(args.output, cubes, points_numbers, cube_positions, args.scale, args.cube_size, args.rho)


if __name__ == "__main__":
    main()
