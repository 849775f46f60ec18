"""One optimisation step of the reference's train_hyper.py (174-214) on MI355X, data-parallel over RCCL.

    trainer = Trainer(weights, alpha=0.75, beta=3.0, lr=1e-5)
    terms = trainer.step(x)              # x [B,64,64,64,1] occupancy, float32 (torch cuda / numpy)

forward   y = A(x); z = HE(y); z~ = z + U(-.5,.5); (loc, s) = HD(z~); scale = max(|s|, lower_bound);
          y~ = y + U(-.5,.5); x~ = S(y~)                                    train_hyper.py:184-191
loss      alpha*(beta*BCE_empty + BCE_full) + delta*bpp_y + gamma*bpp_z         train_hyper.py:193-199, loss.py:8-33
backward  explicit reverse pass over the layer tables (the reference uses tf.GradientTape, 202-207)
update    tf.train.AdamOptimizer defaults in its TF1 form (104, 209-214)
DP        every rank runs the reference's batch on its own GPU; the flat gradient buffer (658 449 floats,
          2.6 MB) is summed with ONE all_reduce per step (loss coefficients are pre-divided by world size, so the
          sum is the mean of the replica gradients); the reference is single-GPU.

Host orchestration only: every tensor operation is a libpcgc_hip.so kernel (layer-level forward, the
gradient kernels of csrc/train.hip).  The forward here is the plain layer-by-layer graph (every activation is
kept for the backward pass), not the fused inference executor.  Gradients match the CPU oracle (oracle/train.py, torch
autograd); bwd-data runs the forward tile kernels on the adjoint filters, bwd-weight the LDS-tiled
register-blocked kernel of csrc/train_dw.hip.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .models import spec

LN2 = float(np.log(2.0))
_DEBUG_HELD = __import__("os").environ.get("PCGC_DEBUG_HELD", "0") == "1"


class _TrainLayer(ctypes.Structure):            # pcgc_train_layer (include/pcgc.h)
    _fields_ = [("kernel", ctypes.c_void_p), ("dkernel", ctypes.c_void_p), ("dbias", ctypes.c_void_p), ("Cin", ctypes.c_int),
                ("Cout", ctypes.c_int), ("ksize", ctypes.c_int), ("stride", ctypes.c_int), ("transposed", ctypes.c_int)]
EB_NAMES = ["%s_%d" % (k, i) for i in range(4) for k in ("matrix", "bais", "factor")]


def _flat_names(nets=None):
    names = []
    for net, layers in (nets or spec.NETS).items():
        for l in layers():
            names.append(("%s/%s/kernel" % (net, l.name), spec.kernel_shape(l)))
            if l.bias:
                names.append(("%s/%s/bias" % (net, l.name), (l.cout,)))
    return names


class Trainer(object):
    def __init__(self, weights, alpha=0.75, beta=3.0, gamma=1.0, delta=1.0, lr=1e-5, lower_bound=1e-9, group=None, nets=None, q4=None):
        """q4 (default: on, PCGC_TRAIN_Q4=0 switches it off): at cube size 64 the 16-channel tensors of the 64^3 stage (and
        the blocks' 8-channel gradients) are kept in the Q4 layout [b][d][h][C/4][w][4] of the inference path instead of
        NDHWC — the row kernels then move 1 KiB per wave instruction; same sums, same gradients."""
        import os
        self.q4 = (os.environ.get("PCGC_TRAIN_Q4", "1") != "0") if q4 is None else bool(q4)
        self._q4_active = None
        self.dev = _lib.require_gpu()
        self.nets = nets or spec.NETS           # layer tables of the trained sub-models (train_factorized.py passes two)
        self.alpha, self.beta, self.gamma, self.delta = float(alpha), float(beta), float(gamma), float(delta)
        self.lr, self.lower_bound, self.group = float(lr), float(lower_bound), group
        self.b1, self.b2, self.eps, self.t = 0.9, 0.999, 1e-8, 0
        # one flat buffer for parameters, one for gradients (single all_reduce), views per variable
        eb_C = int(weights["estimator/matrix_0"].shape[0])
        self.eb_C = eb_C
        entries = _flat_names(self.nets) + [("estimator/" + n, tuple(weights["estimator/" + n].shape)) for n in EB_NAMES]
        total = sum(int(np.prod(s)) for _, s in entries)
        self.flat_p = torch.empty(total, dtype=torch.float32, device=self.dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=self.dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=self.dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=self.dev)
        self.p, self.g, off = {}, {}, 0
        for name, shape in entries:
            n = int(np.prod(shape))
            self.p[name] = self.flat_p[off:off + n].view(*shape)
            self.g[name] = self.flat_g[off:off + n].view(*shape)
            self.p[name].copy_(torch.from_numpy(np.ascontiguousarray(weights[name], np.float32)))
            off += n
        self.eb_off = total - eb_C * 44                 # the 12 estimator tensors are contiguous at the end
        self.fused_vrn = True                           # VRN blocks through pcgc_vrn_fwd_train where a fused kernel exists
        # library-side plan over every conv layer: filters packed / flipped for all layers in two launches per step and
        # all weight-gradient reductions in one (csrc/train_plan.hip)
        self._layer_index, descs = {}, []
        for net in self.nets:
            for l in self.nets[net]():
                self._layer_index[(net, l.name)] = len(descs)
                descs.append((net, l))
        arr = (_TrainLayer * len(descs))()
        for i, (net, l) in enumerate(descs):
            arr[i].kernel = self.p["%s/%s/kernel" % (net, l.name)].data_ptr()
            arr[i].dkernel = self.g["%s/%s/kernel" % (net, l.name)].data_ptr()
            arr[i].dbias = self.g["%s/%s/bias" % (net, l.name)].data_ptr() if l.bias else None
            arr[i].Cin, arr[i].Cout, arr[i].ksize = l.cin, l.cout, l.k
            arr[i].stride, arr[i].transposed = (2, 1) if l.kind == "tconv" else (l.stride, 0)
        plan = ctypes.c_void_p()
        _lib.check(_lib.hip().pcgc_train_plan_create(ctypes.cast(arr, ctypes.c_void_p), len(descs), ctypes.byref(plan)), "pcgc_train_plan_create")
        self._plan = plan
        # the small launches of the 16^3 stage's weight gradients run at the end of the reverse pass, equal shapes together
        # (PCGC_TRAIN_DEFER_DW=0: every layer's own launch, as before); their operands are held until then (self._held)
        self._defer = os.environ.get("PCGC_TRAIN_DEFER_DW", "1") != "0"
        self._held = []
        _lib.check(_lib.hip().pcgc_train_plan_defer_small(plan, int(self._defer)), "pcgc_train_plan_defer_small")
        # The weight gradients of the 64^3 / 32^3 stages are leaves of the reverse pass: they go to a second stream and run
        # NEXT TO the chain of bwd-data kernels (each alone keeps the matrix pipe 60-65 % busy: LDS and registers leave room
        # for the other's waves on the same CU).  Same kernels, same partial sums, same final reduction: bit-identical.  Their
        # operands are held until the streams have joined (self._held), as for the deferred launches.  PCGC_TRAIN_DW_STREAM=0: off.
        self._dw_stream = torch.cuda.Stream(device=self.dev) if os.environ.get("PCGC_TRAIN_DW_STREAM", "1") != "0" else None

    def __del__(self):
        plan, self._plan = getattr(self, "_plan", None), None
        if plan:
            try:
                _lib.hip().pcgc_train_plan_destroy(plan)
            except Exception:            # interpreter shutdown
                pass

    @staticmethod
    def _q4_flags(net, l):
        """(x_q4, y_q4) of a layer when the 64^3 stage is kept in Q4: the stage's boundary layers and, inside its C = 16
        blocks, conv1_1 / conv2_1 (they read the 16-channel block input) and conv1_2 / conv2_3 (their output gradients are
        the 8-channel halves of the block's)."""
        if net not in ("analysis_transform", "synthesis_transform"):
            return 0, 0
        if l.name in ("conv_in", "up_2"):
            return 0, 1
        if l.name in ("down_1", "deconv_out"):
            return 1, 0
        if "/" in l.name and (l.cin, l.cout) == (16, 4):
            return 1, 0
        if "/" in l.name and (l.cin, l.cout) == (4, 8):
            return 0, 1
        return 0, 0

    def _set_layout(self, D):
        """Q4 is a property of the 64^3 stage: on for cube size 64 (where every kernel of the stage has a Q4 form), off
        otherwise; told to the plan layer by layer whenever it changes."""
        if getattr(self, "_has_q4_stage", None) is None:                    # once: the step runs this every time
            names = {l.name for net in self.nets for l in self.nets[net]()}
            self._has_q4_stage = {"conv_in", "down_1", "up_2", "deconv_out"} <= names
        # (fused_vrn = False is the layer-by-layer cross-check path of the tests: its generic kernels read NDHWC only)
        active = bool(self.q4 and self.fused_vrn and D == 64 and self._has_q4_stage)
        if active != self._q4_active:
            for (net, name), li in self._layer_index.items():
                l = next(l_ for l_ in self.nets[net]() if l_.name == name)
                xq, yq = self._q4_flags(net, l) if active else (0, 0)
                _lib.check(_lib.hip().pcgc_train_plan_set_layout(self._plan, li, xq, yq), "pcgc_train_plan_set_layout")
            self._q4_active = active
        return active

    def _prepare(self):
        """Start of a pass over the networks: the plan's packed / flipped filters follow the current parameter values."""
        _lib.check(_lib.hip().pcgc_train_plan_prepare(self._plan, _lib.stream()), "pcgc_train_plan_prepare")

    # ------------------------------------------------------------------ helpers
    def weights(self):
        return {k: v.detach().cpu().numpy().copy() for k, v in self.p.items()}

    def _conv(self, net, l, x, x_relu=False):
        """x_relu: x is the output of a ReLU (a relu layer or a VRN block) — then the gradient this layer sends back to x
        is masked by (x > 0) inside the bwd-data kernel and the producer needs no separate ReLU-gradient pass."""
        b = self.p["%s/%s/bias" % (net, l.name)] if l.bias else None
        B, D = int(x.shape[0]), int(x.shape[1])
        dout = 2 * D if l.kind == "tconv" else D // l.stride
        y = torch.empty((B, dout, dout, dout, l.cout), dtype=torch.float32, device=self.dev)
        _lib.check(_lib.hip().pcgc_train_conv_fwd(self._plan, self._layer_index[(net, l.name)], _lib.dptr(x), _lib.dptr(b), _lib.dptr(y),
                                                  B, D, int(l.relu), _lib.stream()), "pcgc_train_conv_fwd")
        return y, (net, l, x, y, bool(x_relu))

    def _conv_pair(self, net, la, lb, xa, xb, xa_relu, xb_relu):
        """Two independent stride-1 layers of a block in one launch (pcgc_train_conv_fwd_pair: the two single calls where no
        pair kernel takes the shapes); returns what two _conv calls would."""
        B, D = int(xa.shape[0]), int(xa.shape[1])
        assert la.kind != "tconv" and lb.kind != "tconv" and la.stride == 1 and lb.stride == 1 and tuple(xa.shape[:4]) == tuple(xb.shape[:4])
        ba = self.p["%s/%s/bias" % (net, la.name)] if la.bias else None
        bb = self.p["%s/%s/bias" % (net, lb.name)] if lb.bias else None
        ya = torch.empty((B, D, D, D, la.cout), dtype=torch.float32, device=self.dev)
        yb = torch.empty((B, D, D, D, lb.cout), dtype=torch.float32, device=self.dev)
        _lib.check(_lib.hip().pcgc_train_conv_fwd_pair(self._plan, self._layer_index[(net, la.name)], self._layer_index[(net, lb.name)],
                                                       _lib.dptr(xa), _lib.dptr(xb), _lib.dptr(ba), _lib.dptr(bb), _lib.dptr(ya), _lib.dptr(yb),
                                                       B, D, int(la.relu), int(lb.relu), _lib.stream()), "pcgc_train_conv_fwd_pair")
        return (ya, (net, la, xa, ya, bool(xa_relu))), (yb, (net, lb, xb, yb, bool(xb_relu)))

    def _conv_bwd_pair(self, ca, cb, dza, dzb):
        """_conv_bwd(ca, dza, premasked=True) and _conv_bwd(cb, dzb, premasked=True) with both bwd-data passes in one launch
        (pcgc_train_conv_bwd_data_pair); the weight gradients as there."""
        self._conv_bwd(ca, dza, premasked=True, need_dx=False)
        self._conv_bwd(cb, dzb, premasked=True, need_dx=False)
        (net, la, xa, _, xa_relu), (_, lb, xb, _, xb_relu) = ca, cb
        dxa, dxb = torch.empty_like(xa), torch.empty_like(xb)
        _lib.check(_lib.hip().pcgc_train_conv_bwd_data_pair(self._plan, self._layer_index[(net, la.name)], self._layer_index[(net, lb.name)],
                                                            _lib.dptr(dza), _lib.dptr(dzb), _lib.dptr(dxa), _lib.dptr(dxb),
                                                            _lib.dptr(xa) if xa_relu else None, _lib.dptr(xb) if xb_relu else None,
                                                            int(xa.shape[0]), int(xa.shape[1]), _lib.stream()), "bwd_data_pair")
        return dxa, dxb

    def _conv_bwd(self, cache, dy, dy_cs=None, dy_co=0, need_dx=True, premasked=False, add_to=None, need_dw=True):
        """dy: gradient w.r.t. this layer's output (channels [dy_co, dy_co + cout) of a dy_cs-channel tensor).
        premasked: dy already carries the layer's own ReLU mask (its consumer fused it).  add_to: gradient of x from its
        other consumers, summed into the result.  Returns dx, masked by (x > 0) when the cache says x is a ReLU output."""
        net, l, x, y, x_relu = cache
        lib = _lib.hip()
        B, D = int(x.shape[0]), int(x.shape[1])
        whole = dy_co == 0 and int(dy_cs or l.cout) == l.cout
        if whole and (premasked or not l.relu):
            dz = dy                                      # nothing to mask, nothing to slice
        else:
            dz = torch.empty_like(y)
            nvox = y.numel() // l.cout
            _lib.check(lib.pcgc_relu_bwd(_lib.dptr(dy), int(dy_cs or l.cout), int(dy_co), _lib.dptr(y) if (l.relu and not premasked) else None,
                                         _lib.dptr(dz), nvox, l.cout, _lib.stream()), "pcgc_relu_bwd")
        li = self._layer_index[(net, l.name)]
        if need_dw:
            _lib.check(lib.pcgc_train_conv_bwd_weight(self._plan, li, _lib.dptr(x), _lib.dptr(dz), B, D, self._dw_launch_stream()), "bwd_weight")
            if (self._defer and D <= 16) or self._dw_stream is not None:
                # The plan launches it in finish_weights and reads x / dz THEN: `_held` keeps them alive, and nothing may
                # write them in between — dz is often the caller's dy itself, so no later _add / add_to= / dpre write may
                # target a tensor that served as the dz of a layer at D <= 16 (true of this step's graph; PCGC_DEBUG_HELD=1
                # records a checksum of both operands here and compares it right before finish_weights).  The same holds for
                # every layer once the weight gradients run on their own stream.
                self._hold(x, dz)
        if not need_dx:
            return None
        dx = add_to if add_to is not None else torch.empty_like(x)
        _lib.check(lib.pcgc_train_conv_bwd_data(self._plan, li, _lib.dptr(dz), _lib.dptr(dx), _lib.dptr(x) if x_relu else None,
                                                _lib.dptr(add_to), B, D, _lib.stream()), "bwd_data")
        return dx

    def _dw_launch_stream(self):
        """The stream a weight-gradient launch goes to: the side stream, told to wait for everything queued on the current
        stream so far (its operands), or the current stream."""
        if self._dw_stream is None:
            return _lib.stream()
        self._dw_stream.wait_stream(torch.cuda.current_stream())
        return ctypes.c_void_p(self._dw_stream.cuda_stream)

    def _join_dw_stream(self):
        if self._dw_stream is not None:
            torch.cuda.current_stream().wait_stream(self._dw_stream)

    def _finish_weights(self):
        """End of the reverse pass: the side stream's weight gradients joined, the deferred ones launched, every final sum."""
        self._join_dw_stream()
        if _DEBUG_HELD:
            self._check_held()
        _lib.check(_lib.hip().pcgc_train_plan_finish_weights(self._plan, _lib.stream()), "pcgc_train_plan_finish_weights")
        self._held.clear()

    def _hold(self, *ts):
        self._held.append((ts, self._held_sum(ts) if _DEBUG_HELD else None))

    @staticmethod
    def _held_sum(ts):
        # bit patterns, not values: order-independent and exact (int64 sums of the int32 views)
        return torch.stack([t.reshape(-1).view(torch.int32).sum(dtype=torch.int64) for t in ts])

    def _check_held(self):
        for i, (ts, want) in enumerate(self._held):
            if want is not None and not torch.equal(self._held_sum(ts), want):
                raise RuntimeError("deferred weight gradient %d: an operand was written between its layer's reverse step and "
                                   "pcgc_train_plan_finish_weights (PCGC_DEBUG_HELD=1)" % i)

    def _add(self, a, b):
        _lib.check(_lib.hip().pcgc_add_inplace(_lib.dptr(a), _lib.dptr(b), a.numel(), _lib.stream()))
        return a

    # ------------------------------------------------------------------ VRN block
    def _vrn(self, net, layers, x, x_relu=True):
        c11, c12, c21, c22, c23 = layers
        C, D = int(x.shape[-1]), int(x.shape[1])
        lib = _lib.hip()
        if self.fused_vrn and lib.pcgc_vrn_fwd_train_supported(D, C):
            # the inference path's v_mfma_f32_4x4x1 row kernels on the training tensors: two launches for the block,
            # keeping every intermediate the reverse pass reads (t12 / t23 only as `pre`, for their sign)
            q = tuple(x.shape[:-1]) + (C // 4,)
            t11, t21, t22 = (torch.empty(q, dtype=torch.float32, device=self.dev) for _ in range(3))
            out = torch.empty_like(x)
            signs = bool(lib.pcgc_vrn_fwd_train_signs_supported(D, C))      # keep only the sign bits of `pre` (all the reverse reads)
            pre = torch.empty(tuple(x.shape[:-1]), dtype=torch.int32, device=self.dev) if signs else torch.empty_like(x)
            ps = []
            for l in layers:
                ps += [self.p["%s/%s/kernel" % (net, l.name)].data_ptr(), self.p["%s/%s/bias" % (net, l.name)].data_ptr()]
            arr = (ctypes.c_void_p * 10)(*ps)
            q4 = bool(self._q4_active and D == 64 and C == 16)
            assert signs or not q4
            fwd = lib.pcgc_vrn_fwd_train_q4 if q4 else (lib.pcgc_vrn_fwd_train_signs if signs else lib.pcgc_vrn_fwd_train)
            _lib.check(fwd(_lib.dptr(x), ctypes.cast(arr, ctypes.c_void_p), _lib.dptr(t11), _lib.dptr(t21), _lib.dptr(t22),
                           _lib.dptr(pre), _lib.dptr(out), int(x.shape[0]), D, C, _lib.stream()), "pcgc_vrn_fwd_train")
            k11, k12 = (net, c11, x, t11, bool(x_relu)), (net, c12, t11, None, True)
            k21, k22, k23 = (net, c21, x, t21, bool(x_relu)), (net, c22, t21, t22, True), (net, c23, t22, None, True)
            return out, ("vrn", out, C, k11, k12, k21, k22, k23, pre)
        # the layers one by one (the 16^3 blocks); independent ones two to a launch where a pair kernel exists
        (t11, k11), (t21, k21) = self._conv_pair(net, c11, c21, x, x, x_relu, x_relu)
        (t12, k12), (t22, k22) = self._conv_pair(net, c12, c22, t11, t21, True, True)
        # conv2_3 and out = relu(x + [t12 | t23]) (one launch where the 1x1x1 layer's tiles can merge what they just wrote)
        B = int(x.shape[0])
        t23 = torch.empty((B, D, D, D, c23.cout), dtype=torch.float32, device=self.dev)
        out = torch.empty_like(x)
        b23 = self.p["%s/%s/bias" % (net, c23.name)] if c23.bias else None
        _lib.check(lib.pcgc_train_conv_fwd_merge(self._plan, self._layer_index[(net, c23.name)], _lib.dptr(t22), _lib.dptr(b23), _lib.dptr(t23),
                                                 int(c23.relu), _lib.dptr(x), _lib.dptr(t12), _lib.dptr(out), C, B, D, _lib.stream()),
                   "pcgc_train_conv_fwd_merge")
        k23 = (net, c23, t22, t23, True)
        return out, ("vrn", out, C, k11, k12, k21, k22, k23, None)

    def _vrn_bwd(self, cache, dout, premasked=False):
        """out = relu(x + [t12 | t23]).  One pass gives dpre = dout * (out > 0) (skipped when the consumer of `out` already
        masked it) and the two path ends' slices masked by t12 > 0 / t23 > 0; everything further down gets its ReLU mask
        from the bwd-data epilogue of the layer above, and the three contributions to dx are summed there too."""
        _, out, C, k11, k12, k21, k22, k23, pre = cache
        nvox = out.numel() // C
        t12, t23 = (pre, None) if pre is not None else (k12[3], k23[3])
        dpre = dout if premasked else torch.empty_like(out)
        half = tuple(out.shape[:-1]) + (C // 2,)
        dz12, dz23 = torch.empty(half, dtype=torch.float32, device=self.dev), torch.empty(half, dtype=torch.float32, device=self.dev)
        lib = _lib.hip()
        D = int(out.shape[1])
        fused_tail = self.fused_vrn and lib.pcgc_vrn_bwd_tail_supported(D, C) and lib.pcgc_vrn_bwd_input_supported(D, C)
        one_pass = fused_tail and premasked and pre is not None and pre.dtype == torch.int32 and bool(lib.pcgc_vrn_bwd_tail_split_supported(D, C))
        q4 = bool(self._q4_active and D == 64 and C == 16)
        if q4 and not one_pass:
            raise _lib.PcgcError("Q4 training layout: the reverse of a 64^3 block must take the one-pass kernel (its consumer masks the gradient)")
        if one_pass:
            pass                                         # the split happens inside pcgc_vrn_bwd_tail_split below
        elif pre is not None and pre.dtype == torch.int32:
            _lib.check(_lib.hip().pcgc_vrn_bwd_split_signs(_lib.dptr(dout), _lib.dptr(out), _lib.dptr(pre),
                                                           None if premasked else _lib.dptr(dpre), _lib.dptr(dz12), _lib.dptr(dz23), nvox, C,
                                                           int(premasked), _lib.stream()), "pcgc_vrn_bwd_split_signs")
        else:
            _lib.check(_lib.hip().pcgc_vrn_bwd_split(_lib.dptr(dout), _lib.dptr(out), _lib.dptr(t12), _lib.dptr(t23),
                                                     None if premasked else _lib.dptr(dpre), _lib.dptr(dz12), _lib.dptr(dz23), nvox, C,
                                                     int(premasked), _lib.stream()), "pcgc_vrn_bwd_split")
        x = k11[2]
        if fused_tail:
            # the three inner layers' bwd-data in one row-kernel pass (dt22 made on the fly for conv2_2^T); their dW as before
            net = k11[0]
            t11, t21, t22 = k12[2], k22[2], k23[2]
            dt11, dt21, dt22 = torch.empty_like(t11), torch.empty_like(t21), torch.empty_like(t22)
            kp = lambda k: self.p["%s/%s/kernel" % (net, k[1].name)].data_ptr()
            if one_pass:                                 # ... and the block tail's reverse (dz12 / dz23 from dout and the sign bits)
                _lib.check((lib.pcgc_vrn_bwd_tail_split_q4 if q4 else lib.pcgc_vrn_bwd_tail_split)(_lib.dptr(dout), _lib.dptr(pre), _lib.dptr(t11), _lib.dptr(t21), _lib.dptr(t22), kp(k12),
                                                       kp(k22), kp(k23), _lib.dptr(dz12), _lib.dptr(dz23), _lib.dptr(dt11), _lib.dptr(dt21),
                                                       _lib.dptr(dt22), int(x.shape[0]), D, C, _lib.stream()), "pcgc_vrn_bwd_tail_split")
            else:
                _lib.check(lib.pcgc_vrn_bwd_tail(_lib.dptr(dz12), _lib.dptr(dz23), _lib.dptr(t11), _lib.dptr(t21), _lib.dptr(t22), kp(k12),
                                                 kp(k22), kp(k23), _lib.dptr(dt11), _lib.dptr(dt21), _lib.dptr(dt22), int(x.shape[0]), D, C,
                                                 _lib.stream()), "pcgc_vrn_bwd_tail")
            self._conv_bwd(k12, dz12, premasked=True, need_dx=False)
            self._conv_bwd(k23, dz23, premasked=True, need_dx=False)
            self._conv_bwd(k22, dt22, premasked=True, need_dx=False)
        else:
            dt11, dt22 = self._conv_bwd_pair(k12, k23, dz12, dz23)      # results masked by t11 > 0 / t22 > 0 in the epilogue
            dt21 = None
        if self.fused_vrn and lib.pcgc_vrn_bwd_input_supported(D, C):
            # both layers that read the block input and the skip connection in one row-kernel pass:
            # dx = (x > 0) * (dpre + conv1_1^T(dt11) + conv2_1^T(dt21)), in place on dpre; their dW as before
            if dt21 is None:
                dt21 = self._conv_bwd(k22, dt22, premasked=True)
            net = k11[0]
            # both layers' dW in one pass over the block input where the fused kernel exists (else the two single calls)
            _lib.check(lib.pcgc_train_conv_bwd_weight_pair(self._plan, self._layer_index[(net, k11[1].name)],
                                                           self._layer_index[(net, k21[1].name)], _lib.dptr(x), _lib.dptr(dt11),
                                                           _lib.dptr(dt21), int(x.shape[0]), D, self._dw_launch_stream()), "bwd_weight_pair")
            if (self._defer and D <= 16) or self._dw_stream is not None:
                self._hold(x, dt11, dt21)
            _lib.check((lib.pcgc_vrn_bwd_input_q4 if q4 else lib.pcgc_vrn_bwd_input)(_lib.dptr(dt11), _lib.dptr(dt21), _lib.dptr(dpre), _lib.dptr(x) if k11[4] else None,
                                              self.p["%s/%s/kernel" % (net, k11[1].name)].data_ptr(),
                                              self.p["%s/%s/kernel" % (net, k21[1].name)].data_ptr(), _lib.dptr(dpre),
                                              int(x.shape[0]), D, C, _lib.stream()), "pcgc_vrn_bwd_input")
            return dpre
        dt21 = self._conv_bwd(k22, dt22, premasked=True)
        net = k11[0]
        _lib.check(lib.pcgc_train_conv_bwd_weight_pair(self._plan, self._layer_index[(net, k11[1].name)],
                                                       self._layer_index[(net, k21[1].name)], _lib.dptr(x), _lib.dptr(dt11),
                                                       _lib.dptr(dt21), int(x.shape[0]), D, self._dw_launch_stream()), "bwd_weight_pair")
        if (self._defer and D <= 16) or self._dw_stream is not None:
            self._hold(x, dt11, dt21)
        # (x > 0) * ((x > 0) * (dpre + conv1_1^T(dt11)) + conv2_1^T(dt21)), in place on dpre: both layers in one launch where the
        # tiles of the 1x1x1 layer can add to what they just wrote, else one after the other
        assert k11[4] == k21[4]
        _lib.check(lib.pcgc_train_conv_bwd_data_chain(self._plan, self._layer_index[(net, k11[1].name)], self._layer_index[(net, k21[1].name)],
                                                      _lib.dptr(dt11), _lib.dptr(dt21), _lib.dptr(dpre), _lib.dptr(x) if k11[4] else None,
                                                      int(x.shape[0]), D, _lib.stream()), "bwd_data_chain")
        return dpre

    # ------------------------------------------------------------------ nets
    def _run_net(self, net, x):
        layers = self.nets[net]()
        caches, i, f, f_relu = [], 0, x, False
        while i < len(layers):
            if layers[i].name.endswith("/conv1_1"):
                f, c = self._vrn(net, layers[i:i + 5], f, f_relu)
                i += 5
                f_relu = True
            else:
                f, c = self._conv(net, layers[i], f, f_relu)
                f_relu = bool(layers[i].relu)
                i += 1
            caches.append(c)
        return f, caches

    def _run_net_bwd(self, caches, dout, need_dx=True):
        d, premasked = dout, False          # premasked: the layer above already applied this layer's ReLU mask to d
        for idx in range(len(caches) - 1, -1, -1):
            c = caches[idx]
            last = idx == 0 and not need_dx
            if c[0] == "vrn":
                d = self._vrn_bwd(c, d, premasked)
                premasked = c[3][4]                      # conv1_1's / conv2_1's x_relu flag = the block input's
            else:
                d = self._conv_bwd(c, d, need_dx=not last, premasked=premasked)
                premasked = c[4]
        return d

    # ------------------------------------------------------------------ one forward/backward
    def forward_backward(self, x, noise_y=None, noise_z=None, grad_scale=1.0, with_iou=False, _before_readback=None):
        """One forward + reverse pass: gradients in self.flat_g, the loss terms returned.  _before_readback(bce_sums): called with
        the BCE sums still on the device right before the step's one read-back — step() queues the optimiser update there, so
        the GPU does not idle while the host waits for six numbers."""
        lib = _lib.hip()
        x = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x, np.float32))
        x = x.to(self.dev, torch.float32).contiguous()
        self._join_dw_stream()                           # a step that raised half-way may have left launches there
        self.flat_g.zero_()
        self._held.clear()
        self._set_layout(int(x.shape[1]))
        self._prepare()
        # ---- forward
        y, ca = self._run_net("analysis_transform", x)
        z, che = self._run_net("hyper_encoder", y)
        if noise_z is None and noise_y is None:              # U(-1/2, 1/2) for both latents from one generator call (two launches, not four)
            r = torch.rand(z.numel() + y.numel(), dtype=torch.float32, device=self.dev).sub_(0.5)
            nz, ny = r[:z.numel()].view_as(z), r[z.numel():].view_as(y)
        else:
            nz = (torch.rand_like(z) - 0.5) if noise_z is None else torch.as_tensor(noise_z, dtype=torch.float32).to(self.dev).contiguous()
            ny = (torch.rand_like(y) - 0.5) if noise_y is None else torch.as_tensor(noise_y, dtype=torch.float32).to(self.dev).contiguous()
        eb_params = self.flat_p[self.eb_off:]
        z_t, lik_z = torch.empty_like(z), torch.empty_like(z)
        _lib.check(lib.pcgc_factorized_likelihood(_lib.dptr(z), _lib.dptr(eb_params), _lib.dptr(nz), _lib.dptr(z_t), _lib.dptr(lik_z),
                                                  z.numel(), self.eb_C, 1e-9, _lib.stream()))
        hd_layers = spec.NETS["hyper_decoder"]()
        f, c1 = self._conv("hyper_decoder", hd_layers[0], z_t)
        f, c2 = self._conv("hyper_decoder", hd_layers[1], f, True)
        f3, c3 = self._conv("hyper_decoder", hd_layers[2], f, True)
        loc, c41 = self._conv("hyper_decoder", hd_layers[3], f3, True)
        s_raw, c42 = self._conv("hyper_decoder", hd_layers[4], f3, True)
        scale = torch.empty_like(s_raw)
        _lib.check(lib.pcgc_abs_max(_lib.dptr(s_raw), self.lower_bound, None, _lib.dptr(scale), s_raw.numel(), _lib.stream()))
        y_t, lik_y = torch.empty_like(y), torch.empty_like(y)
        _lib.check(lib.pcgc_laplace_likelihood(_lib.dptr(y), _lib.dptr(loc), _lib.dptr(scale), _lib.dptr(ny), _lib.dptr(y_t),
                                               _lib.dptr(lik_y), y.numel(), 1e-9, _lib.stream()))
        x_t, cs = self._run_net("synthesis_transform", y_t)
        # ---- loss terms
        # the BCE sums and the two log-likelihood sums: two launches (bit-identical to pcgc_bce_sums + 2 x pcgc_sum_log)
        loss_sums = torch.empty(6, dtype=torch.float64, device=self.dev)
        sums, logs = loss_sums[:4], loss_sums[4:]
        ws = torch.empty(int(lib.pcgc_train_loss_sums_workspace_bytes(x_t.numel())), dtype=torch.uint8, device=self.dev)
        _lib.check(lib.pcgc_train_loss_sums(_lib.dptr(x_t), _lib.dptr(x), x_t.numel(), _lib.dptr(lik_y), lik_y.numel(), _lib.dptr(lik_z),
                                            lik_z.numel(), _lib.dptr(sums), _lib.dptr(logs), _lib.dptr(ws), ws.numel(), _lib.stream()),
                   "pcgc_train_loss_sums")
        # ---- backward.  The loss terms divide by the numbers of empty / occupied voxels (n0, n1 = sums[1], sums[3]); the reverse
        # kernels form those coefficients on the device (pcgc_*_bwd_dev: the same double-precision expressions), so the host reads
        # the four sums and the two log-likelihoods only AFTER the whole reverse pass is queued — no stall in the middle of the step
        gs = float(grad_scale)
        n1_dev = _lib.dptr(sums[3:4])
        dx_t = torch.empty_like(x_t)
        _lib.check(lib.pcgc_bce_bwd_dev(_lib.dptr(x_t), _lib.dptr(x), _lib.dptr(sums), gs * self.alpha * self.beta, gs * self.alpha,
                                        _lib.dptr(dx_t), x_t.numel(), _lib.stream()))
        dy_t = self._run_net_bwd(cs, dx_t)
        dy_l, dloc, dscale = torch.empty_like(y), torch.empty_like(y), torch.empty_like(y)
        _lib.check(lib.pcgc_laplace_likelihood_bwd_dev(_lib.dptr(y_t), _lib.dptr(loc), _lib.dptr(scale), gs * self.delta, -LN2, n1_dev,
                                                       1e-9, _lib.dptr(dy_l), _lib.dptr(dloc), _lib.dptr(dscale), y.numel(), _lib.stream()))
        self._add(dy_t, dy_l)
        ds_raw = torch.empty_like(s_raw)
        _lib.check(lib.pcgc_abs_max(_lib.dptr(s_raw), self.lower_bound, _lib.dptr(dscale), _lib.dptr(ds_raw), s_raw.numel(), _lib.stream()))
        df3 = self._conv_bwd(c42, ds_raw, add_to=self._conv_bwd(c41, dloc))          # both heads, masked by f3 > 0
        dz_t = self._conv_bwd(c1, self._conv_bwd(c2, self._conv_bwd(c3, df3, premasked=True), premasked=True), premasked=True)
        dz_l = torch.empty_like(z)
        wsf = torch.empty(int(lib.pcgc_factorized_bwd_workspace_bytes(self.eb_C)), dtype=torch.uint8, device=self.dev)
        _lib.check(lib.pcgc_factorized_likelihood_bwd_dev(_lib.dptr(z_t), _lib.dptr(eb_params), gs * self.gamma, -LN2, n1_dev, 1e-9,
                                                          _lib.dptr(dz_l), _lib.dptr(self.flat_g[self.eb_off:]), z.numel(), self.eb_C,
                                                          _lib.dptr(wsf), wsf.numel(), _lib.stream()))
        self._add(dz_t, dz_l)
        dy_he = self._run_net_bwd(che, dz_t)
        self._add(dy_t, dy_he)
        self._run_net_bwd(ca, dy_t, need_dx=False)
        self._finish_weights()
        if _before_readback is not None:
            _before_readback(sums)
        s0, n0, s1, n1, ly, lz = (float(v) for v in loss_sums.cpu().numpy())          # the step's one read-back
        if n0 == 0 or n1 == 0:
            # an all-full / all-empty batch: the *_bwd_dev kernels divided by the zero count ON THE DEVICE and the reverse pass
            # carried inf / NaN into flat_g before this read-back could stop the step — leave no poisoned gradients behind
            self.flat_g.zero_()
            raise ZeroDivisionError("train step on a batch with %s voxels: loss.get_bce_loss averages over both classes "
                                    "(loss.py:8-33); the gradients were zeroed" % ("no empty" if n0 == 0 else "no occupied"))
        num_points = n1
        empty, full = s0 / n0, s1 / n1
        bpp_y, bpp_z = ly / (-LN2 * num_points), lz / (-LN2 * num_points)
        loss = self.alpha * (self.beta * empty + full) + self.delta * bpp_y + self.gamma * bpp_z
        terms = dict(loss=loss, bpp_y=bpp_y, bpp_z=bpp_z, empty=empty, full=full, num_points=num_points)
        if with_iou:
            terms["IoU"] = self.iou(x_t, x)
        return terms

    @staticmethod
    def iou(x_t, x):
        """Post-process classification of the training loop (train_hyper.py:216-226): per-cube adaptive top-k at
        rho = 1 (select_voxels), then loss.get_classify_metrics -> IoU."""
        from . import loss as loss_mod
        from .dataprocess import inout_points as iop
        nums = x.reshape(x.shape[0], -1).sum(dim=1).to(torch.int64).cpu().numpy()
        mask = iop.select_voxels(x_t, nums, 1.0)
        return loss_mod.get_classify_metrics(mask.to(torch.float32), x)[2]

    def evaluate(self, x):
        """One batch of the held-out evaluation (train_hyper.py:126-162): the forward pass with training=False —
        rounded latents instead of additive noise — and no gradients.  -> dict(bpp_y, bpp_z, IoU, num_points)."""
        lib = _lib.hip()
        x = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x, np.float32))
        x = x.to(self.dev, torch.float32).contiguous()
        self._set_layout(int(x.shape[1]))
        self._prepare()
        y, _ = self._run_net("analysis_transform", x)
        z, _ = self._run_net("hyper_encoder", y)
        z_t, lik_z = torch.empty_like(z), torch.empty_like(z)
        _lib.check(lib.pcgc_factorized_likelihood(_lib.dptr(z), _lib.dptr(self.flat_p[self.eb_off:]), None, _lib.dptr(z_t), _lib.dptr(lik_z),
                                                  z.numel(), self.eb_C, 1e-9, _lib.stream()))
        hd = spec.NETS["hyper_decoder"]()
        f, _ = self._conv("hyper_decoder", hd[0], z_t)
        f, _ = self._conv("hyper_decoder", hd[1], f, True)
        f3, _ = self._conv("hyper_decoder", hd[2], f, True)
        loc, _ = self._conv("hyper_decoder", hd[3], f3, True)
        s_raw, _ = self._conv("hyper_decoder", hd[4], f3, True)
        scale = torch.empty_like(s_raw)
        _lib.check(lib.pcgc_abs_max(_lib.dptr(s_raw), self.lower_bound, None, _lib.dptr(scale), s_raw.numel(), _lib.stream()))
        y_t, lik_y = torch.empty_like(y), torch.empty_like(y)
        _lib.check(lib.pcgc_laplace_likelihood(_lib.dptr(y), _lib.dptr(loc), _lib.dptr(scale), None, _lib.dptr(y_t),
                                               _lib.dptr(lik_y), y.numel(), 1e-9, _lib.stream()))
        x_t, _ = self._run_net("synthesis_transform", y_t)
        logs = torch.empty(2, dtype=torch.float64, device=self.dev)
        ws2 = torch.empty(int(lib.pcgc_sum_log_workspace_bytes()), dtype=torch.uint8, device=self.dev)
        _lib.check(lib.pcgc_sum_log(_lib.dptr(lik_y), lik_y.numel(), _lib.dptr(logs[0:1]), _lib.dptr(ws2), ws2.numel(), _lib.stream()))
        _lib.check(lib.pcgc_sum_log(_lib.dptr(lik_z), lik_z.numel(), _lib.dptr(logs[1:2]), _lib.dptr(ws2), ws2.numel(), _lib.stream()))
        ly, lz = (float(v) for v in logs.cpu().numpy())
        num_points = float((x.sum(dim=-1) > 0).sum().item())
        return dict(bpp_y=ly / (-LN2 * num_points), bpp_z=lz / (-LN2 * num_points), IoU=self.iou(x_t, x), num_points=num_points)

    # ------------------------------------------------------------------ optimiser step (with DP all-reduce)
    def step(self, x, noise_y=None, noise_z=None, with_iou=False):
        import torch.distributed as dist
        world = dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1
        if world == 1:
            # the update is queued before the loss terms are read back; on a batch without empty / occupied voxels the kernel
            # skips it on the device (pcgc_adam_step_guarded) and forward_backward raises as it always did
            t0 = self.t
            try:
                return self.forward_backward(x, noise_y, noise_z, with_iou=with_iou, _before_readback=self.apply_gradients)
            except ZeroDivisionError:
                self.t = t0
                raise
        terms = self.forward_backward(x, noise_y, noise_z, grad_scale=1.0 / world, with_iou=with_iou)
        dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.group)          # ONE 2.6 MB collective per step
        self.apply_gradients()
        return terms

    # ------------------------------------------------------------------ checkpoint / resume (train_hyper.py:255-284)
    def save(self, ckpt_dir, with_optimizer=True):
        """checkpoint.save(): TensorFlow tensor-bundle files ckpt-<global_step> + `checkpoint` state (pcgcv1_amd/tf_bundle.py);
        the model variables under the reference's key names, global_step, and — with_optimizer — Adam's m / v as optimizer
        slots (the reference puts main_optimizer into its Checkpoint only with --reset_optimizer != 0, train_hyper.py:107-121)."""
        from . import checkpoint
        t = self.weights()
        off = 0
        m, v = (self.flat_m.cpu().numpy(), self.flat_v.cpu().numpy()) if with_optimizer else (None, None)
        for name, p in self.p.items():
            n = p.numel()
            if with_optimizer:
                t[name + "/.OPTIMIZER_SLOT/main_optimizer/m"] = m[off:off + n].reshape(tuple(p.shape))
                t[name + "/.OPTIMIZER_SLOT/main_optimizer/v"] = v[off:off + n].reshape(tuple(p.shape))
            off += n
        t["global_step"] = np.asarray(self.t, np.int64)
        return checkpoint.save_tf(t, ckpt_dir, self.t)

    def restore(self, ckpt_dir, reset_optimizer=False, reset_step=None):
        """Resume from the latest checkpoint of ckpt_dir (get_checkpoint_state + restore, train_hyper.py:275-284).
        reset_optimizer zeroes the Adam slots; reset_step (default: same as reset_optimizer) zeroes the step counter —
        the reference restores global_step whenever it resumes its own run and only assigns 0 on the --init_ckpt_dir
        warm start (281-284), whatever --reset_optimizer says."""
        if reset_step is None:
            reset_step = reset_optimizer
        from . import tf_bundle
        prefix = tf_bundle.latest_checkpoint(ckpt_dir)
        if prefix is None:
            raise FileNotFoundError("no TensorFlow checkpoint under %r" % ckpt_dir)
        raw = tf_bundle.read_bundle(prefix)
        # the model variables bind the way checkpoint.load binds them for the codec (object graph first, alias edges, key
        # names last): a bundle the codec accepts is a bundle the trainer resumes from; slots and global_step go by key name
        from . import checkpoint
        bound = checkpoint.load_prefix(prefix)
        off = 0
        for name, p in self.p.items():
            if name not in bound:
                raise KeyError("%s: variable %r missing" % (prefix, name))
            p.copy_(torch.from_numpy(np.ascontiguousarray(bound[name], np.float32)))
            n = p.numel()
            for slot, flat in (("m", self.flat_m), ("v", self.flat_v)):
                key = name + "/.OPTIMIZER_SLOT/main_optimizer/" + slot
                if reset_optimizer or key not in raw:
                    flat[off:off + n].zero_()
                else:
                    flat[off:off + n].copy_(torch.from_numpy(np.ascontiguousarray(raw[key], np.float32).reshape(-1)))
            off += n
        self.t = 0 if reset_step else int(np.asarray(raw.get("global_step", 0)).reshape(-1)[0])
        return prefix

    def apply_gradients(self, guard=None):
        """guard: the step's BCE sums on the device — the update is then skipped there when the batch had no empty or no
        occupied voxel (the caller finds out at its read-back and undoes the step count)."""
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        if guard is not None:
            _lib.check(_lib.hip().pcgc_adam_step_guarded(_lib.dptr(self.flat_p), _lib.dptr(self.flat_g), _lib.dptr(self.flat_m),
                                                         _lib.dptr(self.flat_v), self.flat_p.numel(), float(lr_t), self.b1, self.b2,
                                                         self.eps, _lib.dptr(guard), _lib.stream()), "pcgc_adam_step_guarded")
            return
        _lib.check(_lib.hip().pcgc_adam_step(_lib.dptr(self.flat_p), _lib.dptr(self.flat_g), _lib.dptr(self.flat_m),
                                             _lib.dptr(self.flat_v), self.flat_p.numel(), float(lr_t), self.b1, self.b2, self.eps,
                                             _lib.stream()), "pcgc_adam_step")


# ---------------------------------------------------------------------------------------------------------------------
# Training driver with the reference's flags (train_hyper.py:31-66, loop 174-268).  The reference samples .h5 point
# files of a fixed dataset path (115-116); here `--data` is a glob of .npy / .ply cube point lists ([n,3] coordinates
# inside a cube_size^3 cube) or "synthetic" for seeded cubes.  One process per GPU under torch.distributed
# (`python -m torch.distributed.run --nproc-per-node 8 -m pcgcv1_amd.train_hyper ...`): every rank draws its own
# batch, gradients are averaged with one all_reduce per step, rank 0 writes the TF-format checkpoints.
# ---------------------------------------------------------------------------------------------------------------------
RATIO_EVAL = 9                 # train_hyper.py:79 — the first 1/9 of the file list is held out for evaluation


def load_cube_points(path):
    """[n,3] integer coordinates of one training cube: .h5 in the schema generate_dataset.py:27-29 writes (dataset
    'data', uint8 [n,3]; through h5py when installed, else dataprocess/h5min.py), .npy, or .ply."""
    if path.endswith(".h5"):
        try:
            import h5py
        except ImportError:
            # h5py is not part of this image: the files the reference's generate_dataset.py writes (one contiguous uint8
            # dataset 'data' in an "earliest"-format file) are read by the minimal pure-Python reader instead
            from .dataprocess import h5min
            return h5min.read_dataset(path, "data").astype(np.int64).reshape(-1, 3)
        with h5py.File(path, "r") as h:
            return h["data"][:].astype(np.int64)
    if path.endswith(".npy"):
        return np.asarray(np.load(path), np.int64).reshape(-1, 3)
    from .dataprocess import inout_points as iop
    return np.asarray(iop.load_ply_data(path), np.int64).reshape(-1, 3)


def _load_cube(path, cube_size):
    pts = load_cube_points(path)
    vol = np.zeros((cube_size,) * 3 + (1,), np.float32)
    ok = np.all((pts >= 0) & (pts < cube_size), axis=1)
    vol[pts[ok, 0], pts[ok, 1], pts[ok, 2], 0] = 1.0
    return vol


def split_file_list(files, ratio=RATIO_EVAL):
    """train_hyper.py:167, 257: (held-out, training) = (files[:n // ratio], files[n // ratio:])."""
    n = len(files) // ratio
    return files[:n], files[n:]


def evaluate_files(tr, files, cube_size, batch_size=8):
    """train_hyper.py:126-162: mean bpp_ae, bpp_hyper and IoU over len(files) // batch_size full batches."""
    nb = len(files) // batch_size
    if nb == 0:
        return None
    acc = {"bpp_y": 0.0, "bpp_z": 0.0, "IoU": 0.0}
    for i in range(nb):
        x = np.stack([_load_cube(f, cube_size) for f in files[i * batch_size:(i + 1) * batch_size]])
        t = tr.evaluate(x)
        for k in acc:
            acc[k] += t[k]
    return {k: v / nb for k, v in acc.items()}


class Summaries(object):
    """The scalars the reference hands to tf.contrib.summary (train_hyper.py:240-244, 262-266) — bpp_ae, bpp_hyper,
    bpp, IoU per display / evaluation step — as one JSON object per line in <log_dir>/scalars.jsonl."""

    def __init__(self, log_dir):
        import os
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "scalars.jsonl")

    def write(self, step, bpp_ae, bpp_hyper, iou):
        import json
        with open(self.path, "a") as f:
            f.write(json.dumps({"step": int(step), "bpp_ae": float(bpp_ae), "bpp_hyper": float(bpp_hyper),
                                "bpp": float(bpp_ae + bpp_hyper), "IoU": float(iou)}) + "\n")


def main(argv=None):
    import argparse
    import glob
    import os
    import time
    import torch.distributed as dist
    from . import checkpoint, synthetic
    ap = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument("--alpha", type=float, default=2.0, help="weights for distoration.")
    ap.add_argument("--beta", type=float, default=3.0, help="Weight for empty position.")
    ap.add_argument("--gamma", type=float, default=1.0, help="Weight for hyper likelihoods.")
    ap.add_argument("--delta", type=float, default=1.0, help="Weight for latent likelihoods.")
    ap.add_argument("--lr", type=float, default=1e-5)
    ap.add_argument("--num_iteration", type=int, default=int(3e5))
    ap.add_argument("--prefix", type=str, default="")
    ap.add_argument("--init_ckpt_dir", type=str, default="")
    ap.add_argument("--reset_optimizer", type=int, default=0)
    ap.add_argument("--lower_bound", type=float, default=1e-9)
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--data", type=str, default="synthetic", help="glob of .npy/.ply cube point lists, or 'synthetic'")
    ap.add_argument("--cube_size", type=int, default=64)
    ap.add_argument("--save_step", type=int, default=5000)
    ap.add_argument("--display_step", type=int, default=100)
    a = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl")
    ckpt_dir = "./checkpoints/%shyper|a%.2fb%.2f/" % (a.prefix, a.alpha, a.beta)          # train_hyper.py:271-272 (the '|' is the reference's)
    # --reset_optimizer as the reference means it (train_hyper.py:107-121): 0 = the optimizer is NOT part of the checkpoint,
    # so every resumed run starts Adam from zero slots; non-zero = main_optimizer is saved and restored with the model
    with_opt = bool(a.reset_optimizer)
    from . import tf_bundle
    if tf_bundle.latest_checkpoint(ckpt_dir):                                              # resume (275-280)
        weights, resume, reset = checkpoint.load(ckpt_dir), ckpt_dir, False
    elif a.init_ckpt_dir:                                                                   # warm start, step 0 (281-284)
        weights, resume, reset = checkpoint.load(a.init_ckpt_dir), (a.init_ckpt_dir if tf_bundle.latest_checkpoint(a.init_ckpt_dir) else None), True
    else:
        weights, resume, reset = synthetic.make_weights(seed=0, profile="dense"), None, True
    tr = Trainer(weights, alpha=a.alpha, beta=a.beta, gamma=a.gamma, delta=a.delta, lr=a.lr, lower_bound=a.lower_bound)
    if resume:
        # the warm start too restores main_optimizer when the file holds it (--reset_optimizer=1 keeps it in the Checkpoint,
        # train_hyper.py:107-121) and only assigns global_step = 0 (281-284): the slots are NOT zeroed by --init_ckpt_dir
        tr.restore(resume, reset_optimizer=not with_opt, reset_step=reset)
    files = [] if a.data == "synthetic" else sorted(glob.glob(a.data))
    if a.data != "synthetic" and not files:
        raise SystemExit("--data %r matches no file" % a.data)
    eval_files, train_files = split_file_list(files)
    if files and not eval_files:
        train_files = files                               # fewer than RATIO_EVAL files: nothing to hold out
    log_dir = "./logs/%shyper_a%.2fb%.2f/" % (a.prefix, a.alpha, a.beta)                   # train_hyper.py:289-296
    eval_log_dir = "./logs/%shyper_eval_a%.2fb%.2f/" % (a.prefix, a.alpha, a.beta)
    writer, eval_writer = (Summaries(log_dir), Summaries(eval_log_dir)) if rank == 0 else (None, None)
    rng = np.random.default_rng([1234 + rank, tr.t])     # a resumed run does not replay the samples it already saw
    eval_rng = np.random.default_rng(3)
    t0, acc, n_acc = time.time(), {}, 0
    import gc
    gc.collect()
    gc.freeze()          # weights, layer tables and the plan live for the whole run: keep full collections off them (a full
                         # collection over the set-up's objects is 50 ms, four 8-cube steps)
    while tr.t < a.num_iteration:
        if files:
            x = np.stack([_load_cube(train_files[i], a.cube_size) for i in rng.choice(len(train_files), a.batch_size,
                                                                                      replace=len(train_files) < a.batch_size)])
        else:
            x = synthetic.make_cubes(seed=int(rng.integers(1 << 30)), n_cubes=a.batch_size, cube_size=a.cube_size)
        terms = tr.step(x, with_iou=True)                     # IoU of every step is averaged into the summaries (240-244)
        for k in ("loss", "bpp_y", "bpp_z", "empty", "full", "IoU"):
            acc[k] = acc.get(k, 0.0) + terms[k]
        n_acc += 1
        if tr.t % a.display_step == 0 and rank == 0:
            print("Iteration:%d  " % tr.t + "  ".join("%s %.4f" % (k, v / n_acc) for k, v in acc.items())
                  + "  (%.1f min)" % ((time.time() - t0) / 60.0), flush=True)
            writer.write(tr.t, acc["bpp_y"] / n_acc, acc["bpp_z"] / n_acc, acc["IoU"] / n_acc)
        if tr.t % a.display_step == 0:
            acc, n_acc = {}, 0
        if tr.t % a.save_step == 0 and rank == 0:
            if eval_files or not files:                       # held-out evaluation before every checkpoint (255-266)
                print("evaluating...", flush=True)
                if files:
                    pick = [eval_files[i] for i in eval_rng.choice(len(eval_files), min(256, len(eval_files)), replace=False)]
                    ev = evaluate_files(tr, pick, a.cube_size, a.batch_size)
                else:
                    ev = tr.evaluate(synthetic.make_cubes(seed=3, n_cubes=a.batch_size, cube_size=a.cube_size))
                if ev:
                    print("Bpps: %.4f + %.4f\nIoU: %.4f" % (ev["bpp_y"], ev["bpp_z"], ev["IoU"]), flush=True)
                    eval_writer.write(tr.t, ev["bpp_y"], ev["bpp_z"], ev["IoU"])
            tr.save(ckpt_dir, with_optimizer=with_opt)
    if rank == 0:
        tr.save(ckpt_dir, with_optimizer=with_opt)
    if world > 1:
        dist.barrier()


if __name__ == "__main__":
    main()
