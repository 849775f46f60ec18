"""The reference's ablation / RD driver (eval_ablation_studies.py:252-345): per input ply, write (or read) the default
rate-point config <rootdir>/cfg/<name>.ini, run every rate section — compress, container, decompress, rho search for the
best D1 / D2 (written back into the .ini), the three reconstructions and their pc_error tables — and write
<rootdir>/csv/<name>.csv.  Same flags as the reference, both modes (hyper: R1 … R7 under checkpoints/hyper, factorized:
R1 … R6 under checkpoints/factorized, eval_ablation_studies.py:54-77); the png plot is not reproduced (the csv holds every
plotted series).
"""
import os

from . import eval as rd


def eval(input_file, rootdir, resolution, mode, cube_size, modelname, fixed_thres, postfix, ckpt_root=None):
    csv_rootdir = os.path.join(rootdir, "csv")
    cfg_rootdir = os.path.join(rootdir, "cfg")
    os.makedirs(csv_rootdir, exist_ok=True)
    _, config_file = rd.set_default_config(input_file, cfg_rootdir, resolution, mode, cube_size, ckpt_root=ckpt_root, modelname=modelname)
    return rd.eval(input_file, csv_rootdir, config_file, resolution, mode=mode, cube_size=cube_size, modelname=modelname,
                   fixed_thres=fixed_thres, postfix=postfix)


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)          # eval_ablation_studies.py:324-345
    ap.add_argument("--input", type=str, nargs="+", default="", dest="input")
    ap.add_argument("--rootdir", type=str, default="./results/", dest="rootdir")
    ap.add_argument("--resolution", type=int, default=1024, dest="resolution")
    ap.add_argument("--mode", type=str, default="hyper", dest="mode")
    ap.add_argument("--cube_size", type=int, default=64, dest="cube_size")
    ap.add_argument("--modelname", type=str, default="models.model_voxception", dest="modelname")
    ap.add_argument("--fixed_thres", type=float, default=None, dest="fixed_thres")
    ap.add_argument("--postfix", type=str, default="", dest="postfix")
    ap.add_argument("--ckpt_root", type=str, default=None, help="where the default config looks for a<alpha>b3 directories "
                                                                "(default ./checkpoints/<mode>)")
    a = ap.parse_args(argv)
    for input_file in sorted(a.input):
        for row in eval(input_file, a.rootdir, a.resolution, a.mode, a.cube_size, a.modelname, a.fixed_thres, a.postfix, a.ckpt_root):
            print(row)


if __name__ == "__main__":
    main()
