"""preprocess / postprocess — same entry points as the reference's process.py (16-52, 54-82).

preprocess: optional scaling -> partition into cubes -> voxelise; postprocess: adaptive top-k
classification -> points -> merge -> optional inverse scaling -> ASCII ply.  Cubes stay in HBM as a
float32 torch tensor [B,cs,cs,cs,1] (pass device=False for the reference's float64 numpy array).
"""
import time

import numpy as np

from .dataprocess import inout_points as iop


def preprocess_points(points, scale, cube_size, min_num, device=True, block=None):
    """block = (rank, world): voxelise (and upload) only that rank's contiguous block of the key-sorted cube list
    (sharding.shard_range) — cubes and points_numbers then cover the block, cube_positions still the whole cloud."""
    points = np.asarray(points)
    if scale != 1:
        down = np.round(points.astype("float32") * scale)
        points = np.unique(down, axis=0).astype(np.int32)       # process.py:29-30 (ply round trip keeps integers)
    points = np.ascontiguousarray(points, np.int32)
    pos, spos, cop = iop.partition(points, cube_size, min_num)
    lo, hi = 0, len(pos)
    if block is not None:
        from .sharding import shard_range
        lo, hi = shard_range(len(pos), int(block[0]), int(block[1]))
    if device:
        cubes = iop.voxelize_partition(points, cop, lo, hi, cube_size)
        points_numbers = cubes.sum(dim=(1, 2, 3, 4)).cpu().numpy().astype(np.uint16)
    else:
        keep = (cop >= lo) & (cop < hi)
        cubes = iop.voxelize(cop[keep] - lo, points[keep] % cube_size, hi - lo, cube_size, device=False)
        points_numbers = np.sum(cubes, axis=(1, 2, 3, 4)).astype(np.uint16)
    return cubes, pos, points_numbers


def preprocess(input_file, scale, cube_size, min_num, device=True, verbose=True, block=None):
    """-> (cubes [B,cs,cs,cs,1], cube_positions [B,3] in first-appearance order, points_numbers uint16 [B])."""
    if verbose:
        print('===== Preprocess =====')
    start = time.time()
    pts = iop.load_ply_data(input_file)
    cubes, pos, nums = preprocess_points(pts, scale, cube_size, min_num, device, block)
    if verbose:
        print("Scaling + Partition + Voxelization: {}s".format(round(time.time() - start, 4)))
        print('cubes shape: {}'.format(tuple(cubes.shape)))
        print('points numbers (sum/mean/max/min): {} {} {} {}'.format(nums.astype(np.int64).sum(), round(nums.mean()),
                                                                      nums.max(), nums.min()))
    return cubes, pos, nums


def postprocess_points(cubes, points_numbers, cube_positions, scale, cube_size, rho, fixed_thres=None):
    mask = iop.select_voxels(cubes, points_numbers, rho, fixed_thres=fixed_thres)
    pts = iop.voxels2merged_points(mask, cube_positions, cube_size)
    if scale == 1:
        return pts
    return pts.astype(np.int32).astype("float32") * float(1 / scale)          # process.py:76-77


def postprocess_masks(output_file, masks, cube_positions, scale, cube_size, verbose=True):
    """Tail of postprocess for occupancy masks that were already classified on the GPUs that decoded them (the sharded
    decoder gathers bit-packed masks, not logits): voxels2points, merge by cube position, scale back, write the ply."""
    import torch
    pts = (iop.voxels2merged_points(masks, cube_positions, cube_size) if torch.is_tensor(masks)
           else iop.merge_points(iop.voxels2points(masks), cube_positions, cube_size))
    if scale != 1:
        pts = pts.astype(np.int32).astype("float32") * float(1 / scale)          # process.py:76-77
    iop.write_ply_data(output_file, pts)
    if verbose:
        print("Write point cloud to {} ({} points)".format(output_file, len(pts)))


def _pwrite_all(fd, data, off):
    import os
    view = memoryview(data)
    while len(view):
        w = os.pwrite(fd, view, off)
        off += w
        view = view[w:]
    return off


class StreamedPostprocess(object):
    """postprocess (process.py:54-82) slice by slice while the decoder is still running: pass the object as
    `on_slice` to transform.decompress_hyper, then call finish().  Every finished slice is classified (top-k), turned into
    points and formatted on a worker thread with a stream of its own while the GPU synthesises the following slices; finish()
    writes the texts in cube order as they become ready.  The file is byte for byte what postprocess writes (same points,
    same order).  scale must be 1 (the float path formats the whole cloud at once: use postprocess)."""

    def __init__(self, output_file, points_numbers, cube_positions, scale, cube_size, rho, fixed_thres=None):
        import torch
        from . import _lib
        if scale != 1:
            raise ValueError("StreamedPostprocess: scale must be 1")
        self.output_file, self.cube_size, self.rho, self.fixed_thres = output_file, int(cube_size), rho, fixed_thres
        self.nums = np.asarray(points_numbers).reshape(-1)
        self.spos = torch.from_numpy(np.ascontiguousarray(iop.ordered_positions(cube_positions), np.int64)).to(_lib.require_gpu())
        self.parts = {}
        self._lib = _lib
        self._nth = {}                                  # slices seen per decoder pipeline (its stream)

    def __call__(self, lo, hi, x):
        """on the decoder's pipeline thread, with its stream current: mark the point in the stream, hand the rest over"""
        import torch
        ev = torch.cuda.Event()
        ev.record()
        # persistent tail streams (the caching allocator keeps a pool per stream: a fresh stream per call would pay hipMalloc — a
        # device-wide wait — for every slice), ONE PER SLICE of a decoder pipeline: torch.nonzero waits for its whole stream, and
        # on a stream shared by the pipeline's slices that included the waits for LATER slices' synthesis already queued behind
        # it — every slice's points then arrived with the last one (tools/exp/t_cli_timeline.py: 20 ms in voxels2merged_points)
        cur = torch.cuda.current_stream()
        nth = self._nth.get(int(cur.cuda_stream), 0)
        self._nth[int(cur.cuda_stream)] = nth + 1
        st = self._lib.side_stream("tail%d" % (nth % 4), cur)
        self.parts[lo] = (hi, self._lib.workers("job").submit(self._slice, lo, hi, x, ev, st))

    def _slice(self, lo, hi, x, ev, st):
        import torch
        torch.cuda.set_device(st.device)
        with torch.cuda.stream(st):
            st.wait_event(ev)
            mask = iop.select_voxels(x, self.nums[lo:hi], self.rho, fixed_thres=self.fixed_thres)
            pts = iop.voxels2merged_points(mask, self.spos[lo:hi], self.cube_size, ordered=True)
        _, body = iop._ply_parts(pts)
        return len(pts), bytes(body)                  # the formatter's buffer belongs to this worker thread: copy out

    def finish(self, verbose=True):
        """-> number of points written"""
        import os
        start = time.time()
        expect = int(sum(int(self.rho * np.array(n)) for n in self.nums))
        head = iop.ply_header(expect)
        fd = os.open(self.output_file, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666)
        try:
            off, total, at, bodies = len(head), 0, 0, []
            for lo in sorted(self.parts):
                hi, fut = self.parts[lo]
                if lo != at:
                    raise RuntimeError("StreamedPostprocess: cubes %d..%d never arrived" % (at, lo))
                n, body = fut.result()
                bodies.append(body)
                off = _pwrite_all(fd, body, off)                          # at its final offset, assuming `expect` points
                total, at = total + n, hi
            if at != len(self.nums):
                raise RuntimeError("StreamedPostprocess: %d of %d cubes arrived" % (at, len(self.nums)))
            real = iop.ply_header(total)
            if len(real) != len(head):                                    # ties moved the count across a power of ten
                os.ftruncate(fd, 0)
                _pwrite_all(fd, real + b"".join(bodies), 0)
            else:
                _pwrite_all(fd, real, 0)
        except BaseException:
            os.close(fd)
            fd = None
            try:
                os.unlink(self.output_file)                               # never leave a file with a hole where the header goes
            except OSError:
                pass
            raise
        finally:
            if fd is not None:
                os.close(fd)
            self.parts = {}
        if verbose:
            print("Classify, extract and write {} points to {} (streamed): {}s after the decoder returned".format(
                total, self.output_file, round(time.time() - start, 4)))
        return total


def postprocess(output_file, cubes, points_numbers, cube_positions, scale, cube_size, rho, fixed_thres=None,
                verbose=True):
    if verbose:
        print('===== Post process =====')
    start = time.time()
    pts = postprocess_points(cubes, points_numbers, cube_positions, scale, cube_size, rho, fixed_thres)
    if verbose:
        print("Classify and extract points: {}s".format(round(time.time() - start, 4)))
    start = time.time()
    iop.write_ply_data(output_file, pts)
    if verbose:
        print("Write point cloud to {}: {}s".format(output_file, round(time.time() - start, 4)))
    return
