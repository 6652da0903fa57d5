"""Training step of the refiner network on the MI355X.

Reference: MBConvBlock.forward / EfficientNet.extract_features in train mode (cosypose/models/efficientnet.py:71-98,
:174-190; batch-statistics BatchNorm, momentum 0.01, eps 1e-3), SwishImplementation.backward and drop_connect
(efficientnet_utils.py:37-48, :83-92), PosePredictor.net_forward (models/pose.py:81-87), the train step
(training/train_pose.py:317-331: backward, clip_grad_norm_(0.5), Adam) and DDP's gradient averaging.

Design: fp32, activations NHWC (rows = pixels, so every 1x1 convolution and its two gradients are plain row-major
GEMMs -> the library's own fp32 MFMA kernels, cosy_train_gemm / cosy_wgrad); everything else -- BatchNorm statistics / apply / backward with the Swish fused
in, depthwise forward / data / weight gradients, squeeze-excite scaling and its gradients, pooling, the stem's
im2col, the loss gradient, gradient norm + clip + Adam on flat buffers -- is hand-written HIP in
csrc/kernels_train.hip behind the C ABI.  The whole network is ONE autograd node (`backbone_train`): its backward
fills the gradients of all 340 parameter tensors, so the reference's loop (`loss.backward(); clip; optimizer.step()`)
runs unchanged on top of it; `FlatAdam` is the fused alternative on flat parameter / gradient buffers, and
`allreduce_gradients` averages them across ranks with one RCCL all-reduce.
With 288 GB of HBM per GPU nothing is recomputed: every conv output and activation of the step stays resident
(~0.3 GB per 240x320 crop).
"""
import math

import numpy as np
import torch

from . import arch, _modwatch
from ._lib import lib, check, ptr, stream, require_device

BN_EPS, BN_MOM = arch.BN_EPS, 0.01
DROP_CONNECT_RATE = 0.2          # efficientnet_utils.py: efficientnet('efficientnet-b3') global params


# ---------------------------------------------------------------------------------------------
# thin wrappers over the C ABI (all tensors fp32, contiguous, on the device)
# ---------------------------------------------------------------------------------------------
_ws = {}


def _workspace(dev):
    """Scratch of the reduction / GEMM kernels: one buffer per (device, stream) -- kernels on one stream use it one after the
    other, two streams (or two models stepped concurrently) never share one."""
    key = (torch.device(dev).index, stream())
    w = _ws.get(key)
    if w is None:
        w = _ws[key] = torch.empty(lib().cosy_train_workspace_bytes(), dtype=torch.uint8, device=dev)
    return w


def bn_stats(x, M, C, running_mean=None, running_var=None):
    mean = torch.empty(C, device=x.device); rstd = torch.empty(C, device=x.device)
    check(lib().cosy_bn_train_stats(ptr(x), M, C, BN_EPS, BN_MOM, ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var),
                                    ptr(_workspace(x.device)), stream()))
    return mean, rstd


def bn_apply(x, mean, rstd, gamma, beta, M, C, act, rowscale=None, HW=1, res=None):
    out = torch.empty_like(x)
    check(lib().cosy_bn_train_apply(ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), M, C, act, ptr(rowscale), HW, ptr(res), ptr(out),
                                    stream()))
    return out


def bn_backward(dout, x, mean, rstd, gamma, beta, M, C, act, rowscale=None, HW=1, out=None, cgate=None, cadd=None, cadd_scale=0.0):
    """cgate / cadd (B,C): the incoming gradient is dout * cgate[sample] + cadd[sample] * cadd_scale, formed inside the kernels (HW rows per sample)"""
    dgamma, dbeta = out if out is not None else (torch.empty(C, device=x.device), torch.empty(C, device=x.device))
    dx = torch.empty_like(x); sums = torch.empty(2 * C, device=x.device)
    check(lib().cosy_bn_train_backward_gated(ptr(dout), ptr(cgate), ptr(cadd), cadd_scale, ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), M, C, act,
                                             ptr(rowscale), HW, ptr(dgamma), ptr(dbeta), 0, ptr(dx), ptr(sums), ptr(_workspace(x.device)), stream()))
    return dx, dgamma, dbeta


def _dw_out(H, W, k, s):
    return (H, W) if s == 1 else ((H - 2) // 2 + 1, (W - 2) // 2 + 1)


def dw_forward(x, wt, B, H, W, C, k, s):
    Ho, Wo = _dw_out(H, W, k, s)
    out = torch.empty(B * Ho * Wo, C, device=x.device)
    check(lib().cosy_dw_train_forward(ptr(x), ptr(wt), B, H, W, C, k, s, ptr(out), stream()))
    return out


def dw_backward(x, dy, wt, B, H, W, C, k, s, add=None, out=None):
    """-> (dx (+ add: the skip connection's gradient of a block without expansion, fused into the store), dw (C,1,k,k))"""
    dx = torch.empty(B * H * W, C, device=x.device)
    check(lib().cosy_dw_train_backward_data_add(ptr(dy), ptr(wt), ptr(add), B, H, W, C, k, s, ptr(dx), stream()))
    # the weight gradient lands in the module's (C, 1, k, k) layout -- in `out` (e.g. the parameter's slice of a flat gradient buffer) when given
    dw = out if out is not None else torch.empty(C, 1, k, k, device=x.device)
    assert dw.numel() == C * k * k and dw.is_contiguous()
    check(lib().cosy_dw_train_backward_weight_ex(ptr(x), ptr(dy), B, H, W, C, k, s, ptr(dw), 1, ptr(_workspace(x.device)), stream()))
    return dx, dw


def rows_mean(a, B, HW, C):
    out = torch.empty(B, C, device=a.device)
    check(lib().cosy_rows_mean(ptr(a), B, HW, C, ptr(out), ptr(_workspace(a.device)), stream()))
    return out


def rows_mean_bn(raw, mean, rstd, gamma, beta, B, HW, C):
    """per-sample mean over the pixels of swish(bn(raw)), the activation recomputed on the fly (never stored)"""
    out = torch.empty(B, C, device=raw.device)
    check(lib().cosy_rows_mean_bn(ptr(raw), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), B, HW, C, ptr(out), ptr(_workspace(raw.device)), stream()))
    return out


def rows_dot_bn(a, raw, mean, rstd, gamma, beta, B, HW, C):
    """per-sample sum over the pixels of a * swish(bn(raw))"""
    out = torch.empty(B, C, device=a.device)
    check(lib().cosy_rows_dot_bn(ptr(a), ptr(raw), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), B, HW, C, ptr(out), ptr(_workspace(a.device)), stream()))
    return out


def bn_apply_gated(x, mean, rstd, gamma, beta, M, C, act, cgate, HW):
    """act(bn(x)) * cgate[sample][c] (cgate (B,C), HW rows per sample)"""
    out = torch.empty_like(x)
    check(lib().cosy_bn_train_apply_gated(ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), M, C, act, ptr(cgate), HW, ptr(out), stream()))
    return out


def rows_dot(a, a2, B, HW, C):
    out = torch.empty(B, C, device=a.device)
    check(lib().cosy_rows_dot(ptr(a), ptr(a2), B, HW, C, ptr(out), ptr(_workspace(a.device)), stream()))
    return out


def rows_scale(a, g, B, HW, C, add=None, add_scale=0.0):
    out = torch.empty_like(a)
    check(lib().cosy_rows_scale(ptr(a), ptr(g), ptr(add), add_scale, B, HW, C, ptr(out), stream()))
    return out


def rows_broadcast(v, scale, B, HW, C):
    out = torch.empty(B * HW, C, device=v.device)
    check(lib().cosy_rows_broadcast(ptr(v), scale, B, HW, C, ptr(out), stream()))
    return out


def act_forward(x, kind):
    out = torch.empty_like(x)
    check(lib().cosy_act_forward(ptr(x), x.numel(), kind, ptr(out), stream()))
    return out


def act_backward(x, dy, kind):
    dx = torch.empty_like(x)
    check(lib().cosy_act_backward(ptr(x), ptr(dy), x.numel(), kind, ptr(dx), stream()))
    return dx


SWISH, SIGMOID = 0, 1


def se_forward(pooled, w_reduce, b_reduce, w_expand, b_expand):
    """squeeze-excite FCs of one block, ONE launch: pooled (B,C) -> (h_pre (B,Cse), gate (B,C)); weights in the module's own layout
    (_se_reduce.weight (Cse,C,1,1), _se_expand.weight (C,Cse,1,1)).  efficientnet.py:85-88."""
    B, C = pooled.shape
    cse = b_reduce.numel()
    h_pre = torch.empty(B, cse, device=pooled.device); gate = torch.empty(B, C, device=pooled.device)
    check(lib().cosy_se_train_forward(ptr(pooled), ptr(w_reduce), ptr(b_reduce), ptr(w_expand), ptr(b_expand), B, C, cse, ptr(h_pre), ptr(gate), stream()))
    return h_pre, gate


def se_backward(dgate, gate, h_pre, pooled, w_reduce, w_expand, out=None):
    """gradients of se_forward from dgate (B,C), two launches: -> dpooled (B,C), dw_reduce (Cse,C), db_reduce (Cse), dw_expand (C,Cse),
    db_expand (C); `out` = the four parameter-gradient tensors to write into (contiguous), else fresh ones."""
    B, C = pooled.shape
    cse = h_pre.shape[1]
    dev = pooled.device
    dpooled = torch.empty(B, C, device=dev)
    dw1, db1, dw2, db2 = out if out is not None else (torch.empty(cse, C, device=dev), torch.empty(cse, device=dev),
                                                       torch.empty(C, cse, device=dev), torch.empty(C, device=dev))
    check(lib().cosy_se_train_backward(ptr(dgate), ptr(gate), ptr(h_pre), ptr(pooled), ptr(w_reduce), ptr(w_expand), B, C, cse, ptr(dpooled),
                                       ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), ptr(_workspace(dev)), stream()))
    return dpooled, dw1, db1, dw2, db2


def fc_small_forward(x, w, bias):
    """x (B,C) @ w (J,C)^T + bias for J <= 16 (the pose head), one launch"""
    B, C = x.shape
    y = torch.empty(B, w.shape[0], device=x.device)
    check(lib().cosy_fc_small_forward(ptr(x), ptr(w), ptr(bias), B, C, w.shape[0], ptr(y), stream()))
    return y


def fc_small_backward(dy, x, w, out=None):
    """-> dx (B,C), dw (J,C), db (J)"""
    B, C = x.shape
    J = w.shape[0]
    dx = torch.empty(B, C, device=x.device)
    dw, db = out if out is not None else (torch.empty(J, C, device=x.device), torch.empty(J, device=x.device))
    check(lib().cosy_fc_small_backward(ptr(dy), ptr(x), ptr(w), B, C, J, ptr(dx), ptr(dw), ptr(db), stream()))
    return dx, dw, db


def wgrad(dY, X, out=None):
    """dW (N,K) = dY^T X for dY (M,N), X (M,K): the streaming fp32 MFMA kernel (cosy_wgrad), every shape.  `out`: a contiguous
    destination of N*K floats (e.g. a parameter's slice of a flat gradient buffer)."""
    M, N = dY.shape
    K = X.shape[1]
    if out is None:
        out = torch.empty(N, K, device=dY.device)
    assert out.numel() == N * K and out.is_contiguous()
    check(lib().cosy_wgrad(ptr(dY), ptr(X), M, N, K, ptr(out), ptr(_workspace(dY.device)), stream()))
    return out


def gemm(A, W, w_is_kn=False, add=None, packed=None):
    """A (M,K) @ W^T for W (N,K)  [w_is_kn=False: a 1x1 convolution's forward]  or  A (M,K) @ W for W (K,N)  [w_is_kn=True: its
    data gradient], plus `add` (M,N): the library's own fp32 MFMA GEMM (cosy_train_gemm), no rocBLAS.  packed: a PackedWeights holding
    W in this orientation (packed at the top of the step with every other weight) -- without it the weight is packed by a launch of its own."""
    M, K = A.shape
    N = W.shape[1] if w_is_kn else W.shape[0]
    assert W.shape == ((K, N) if w_is_kn else (N, K)) and A.is_contiguous() and W.is_contiguous()
    out = torch.empty(M, N, device=A.device)
    if packed is not None:
        check(lib().cosy_train_gemm_packed(ptr(A), ptr(packed.pool), packed.offset(W, w_is_kn), M, K, N, ptr(add), ptr(out), stream()))
    else:
        check(lib().cosy_train_gemm(ptr(A), ptr(W), int(w_is_kn), M, K, N, ptr(add), ptr(out), ptr(_workspace(A.device)), stream()))
    return out


class PackedWeights:
    """The fragment-order copies of a set of 1x1-convolution weights, each in both orientations (forward: (N,K); data gradient: the same
    storage read as (K,N)), refreshed by ONE launch (cosy_train_pack_all) -- the weights change once per step, in Adam, not between the
    ~100 GEMMs that read them.  The plan (where each weight lives) is rebuilt only when the set of tensors changes."""
    _cache = {}            # (device, stream, weight set) -> instance; a handful at most (one per model trained in this process)
    MAX_SETS = 4

    @classmethod
    def current(cls, weights):
        """the instance of (device, stream, this set of 2-D fp32 tensors), planned on first sight, packed now.  Every weight set owns its pool:
        a backward never finds its forward's packed weights overwritten by another model's forward."""
        key = (weights[0].device.index, stream()) + tuple((w.data_ptr(), w.shape[0], w.shape[1]) for w in weights)
        me = cls._cache.pop(key, None)
        if me is None:
            me = cls(weights)
            while len(cls._cache) >= cls.MAX_SETS:
                cls._cache.pop(next(iter(cls._cache)))         # least recently used
        cls._cache[key] = me
        check(lib().cosy_train_pack_all(ptr(me.plan), me.plan.shape[0], me.n_blocks, ptr(me.pool), stream()))
        return me

    def __init__(self, weights):
        import ctypes as c
        n = 2 * len(weights)
        W = (c.c_void_p * n)(*[w.data_ptr() for w in weights for _ in (0, 1)])
        # forward: W (N,K) as it is stored; data gradient dX = dY . W: the same storage as a (K', N') = (N, K) matrix, w_is_kn
        K = (c.c_int * n)(*[v for w in weights for v in (w.shape[1], w.shape[0])])
        N = (c.c_int * n)(*[v for w in weights for v in (w.shape[0], w.shape[1])])
        kn = (c.c_int * n)(*[v for _ in weights for v in (0, 1)])
        plan = torch.empty(n, 8, dtype=torch.int64)
        floats, blocks = c.c_longlong(), c.c_longlong()
        check(lib().cosy_train_pack_plan(n, W, K, N, kn, plan.data_ptr(), c.byref(floats), c.byref(blocks)))
        dev = weights[0].device
        self.plan = plan.to(dev)
        self.pool = torch.empty(floats.value, device=dev)
        self.offsets = {(int(plan[e, 0]), bool(e & 1)): int(plan[e, 1]) for e in range(n)}
        self.n_blocks = blocks.value

    def offset(self, W, w_is_kn):
        return self.offsets[(W.data_ptr(), bool(w_is_kn))]


# ---------------------------------------------------------------------------------------------
# the network as one autograd node
# ---------------------------------------------------------------------------------------------
def param_names():
    """Trainable tensors of PosePredictor in named_parameters() order (reference state_dict names)."""
    names = ['backbone._conv_stem.weight', 'backbone._bn0.weight', 'backbone._bn0.bias']
    for i, (k, s, e, cin, cout) in enumerate(arch.B3_BLOCKS):
        p = f'backbone._blocks.{i}.'
        if e != 1:
            names += [p + '_expand_conv.weight', p + '_bn0.weight', p + '_bn0.bias']
        names += [p + '_depthwise_conv.weight', p + '_bn1.weight', p + '_bn1.bias',
                  p + '_se_reduce.weight', p + '_se_reduce.bias', p + '_se_expand.weight', p + '_se_expand.bias',
                  p + '_project_conv.weight', p + '_bn2.weight', p + '_bn2.bias']
    names += ['backbone._conv_head.weight', 'backbone._bn1.weight', 'backbone._bn1.bias', 'pose_fc.weight', 'pose_fc.bias']
    return names


_keep_prob = {}


def make_drop_connect_scales(B, device, rate=DROP_CONNECT_RATE, generator=None):
    """{block index: (B,) mask/keep_prob} as drop_connect draws them (efficientnet_utils.py:83-92): blocks with a skip
    connection only, rate scaled by idx/26 (efficientnet.py:182-185)."""
    out = {}
    if not rate:
        return out
    n = len(arch.B3_BLOCKS)
    ids = [i for i, (k, s, e, cin, cout) in enumerate(arch.B3_BLOCKS) if s == 1 and cin == cout and rate * float(i) / n]
    key = (torch.device(device), float(rate))
    keep = _keep_prob.get(key)
    if keep is None:       # uploaded once: torch.tensor(..., device=) is a synchronous copy that stops the host until the queued kernels have run
        keep = _keep_prob[key] = torch.tensor([1.0 - rate * float(i) / n for i in ids], device=device).unsqueeze(1)
    scales = torch.floor(keep + torch.rand(len(ids), B, device=device, generator=generator)) / keep      # all blocks at once
    for j, i in enumerate(ids):
        out[i] = scales[j]
    return out


class _Net:
    """Forward (saving what the backward needs) and backward of backbone + pooling + pose_fc."""

    def __init__(self, P, buffers, stage=None):
        self.P, self.buf = P, buffers
        self.stage = stage       # name -> destination view of the parameter's gradient (FlatAdam's staging buffer), or None

    def _dst(self, name):
        return self.stage[name] if self.stage is not None else None

    # ---- helpers
    def _bn_f(self, tape, name, raw, M, C, act, rowscale=None, HW=1, res=None):
        rm, rv = self.buf.get(name + '.running_mean'), self.buf.get(name + '.running_var')
        mean, rstd = bn_stats(raw, M, C, rm, rv)
        tape[name] = (raw, mean, rstd, M, C, act, rowscale, HW)
        return bn_apply(raw, mean, rstd, self.P[name + '.weight'], self.P[name + '.bias'], M, C, act, rowscale, HW, res)

    def _bn_b(self, tape, grads, name, dout, cgate=None, cadd=None, cadd_scale=0.0, HWg=None):
        raw, mean, rstd, M, C, act, rowscale, HW = tape[name]
        out = (self._dst(name + '.weight'), self._dst(name + '.bias')) if self.stage is not None else None
        if cgate is not None:
            assert rowscale is None
            HW = HWg
        dx, dg, db = bn_backward(dout, raw, mean, rstd, self.P[name + '.weight'], self.P[name + '.bias'], M, C, act, rowscale, HW, out=out,
                                 cgate=cgate, cadd=cadd, cadd_scale=cadd_scale)
        grads[name + '.weight'], grads[name + '.bias'] = dg, db
        return dx

    def _conv_weights(self):
        """the 1x1-convolution weights as 2-D matrices, in network order"""
        P, out = self.P, []
        for i, (k, s, e, cin, cout) in enumerate(arch.B3_BLOCKS):
            p = f'backbone._blocks.{i}.'
            if e != 1:
                out.append(P[p + '_expand_conv.weight'].view(cin * e, cin))
            out.append(P[p + '_project_conv.weight'].view(cout, cin * e))
        out.append(P['backbone._conv_head.weight'].view(arch.HEAD_C, -1))
        return out

    def forward(self, x8, drop):
        P = self.P
        tape = {}
        B, H, W, _ = x8.shape
        dev = x8.device
        pk = self.packed = PackedWeights.current(self._conv_weights())      # one launch: every weight, both orientations
        # stem: im2col + GEMM, BN, Swish
        Ho, Wo = (H - 2) // 2 + 1, (W - 2) // 2 + 1
        cols = torch.empty(B * Ho * Wo, 56, device=dev)                 # 54 patch values + 2 zero columns: 16-byte aligned rows
        check(lib().cosy_stem_im2col_ld(ptr(x8), B, H, W, 56, ptr(cols), stream()))
        w2d = torch.nn.functional.pad(P['backbone._conv_stem.weight'].permute(0, 2, 3, 1).reshape(arch.STEM_C, 54), (0, 2))
        raw = gemm(cols, w2d)
        tape['stem'] = cols
        x = self._bn_f(tape, 'backbone._bn0', raw, B * Ho * Wo, arch.STEM_C, 1)
        H, W = Ho, Wo
        for i, (k, s, e, cin, cout) in enumerate(arch.B3_BLOCKS):
            p = f'backbone._blocks.{i}.'
            M, cmid = B * H * W, cin * e
            inp = x
            if e != 1:
                raw = gemm(inp, P[p + '_expand_conv.weight'].view(cmid, cin), packed=pk)
                a0 = self._bn_f(tape, p + '_bn0', raw, M, cmid, 1)
            else:
                a0 = inp
            wt = P[p + '_depthwise_conv.weight'].view(cmid, k * k).t().contiguous()
            raw = dw_forward(a0, wt, B, H, W, cmid, k, s)
            Ho, Wo = _dw_out(H, W, k, s)
            Mo, HWo = B * Ho * Wo, Ho * Wo
            # BatchNorm 1 + Swish + squeeze-excite without the unscaled activation a1 in memory: statistics, then the per-sample means of
            # swish(bn(raw)) recomputed on the fly, the gate, and ONE pass that writes swish(bn(raw)) * gate (the project conv's input)
            n1 = p + '_bn1'
            mean1, rstd1 = bn_stats(raw, Mo, cmid, self.buf.get(n1 + '.running_mean'), self.buf.get(n1 + '.running_var'))
            tape[n1] = (raw, mean1, rstd1, Mo, cmid, 1, None, 1)
            g1, b1 = P[n1 + '.weight'], P[n1 + '.bias']
            pooled = rows_mean_bn(raw, mean1, rstd1, g1, b1, B, HWo, cmid)
            h_pre, g = se_forward(pooled, P[p + '_se_reduce.weight'], P[p + '_se_reduce.bias'], P[p + '_se_expand.weight'], P[p + '_se_expand.bias'])
            a2 = bn_apply_gated(raw, mean1, rstd1, g1, b1, Mo, cmid, 1, g, HWo)
            raw = gemm(a2, P[p + '_project_conv.weight'].view(cout, cmid), packed=pk)
            skip = s == 1 and cin == cout
            rowscale = drop.get(i) if (skip and drop) else None
            x = self._bn_f(tape, p + '_bn2', raw, Mo, cout, 0, rowscale, HWo, inp if skip else None)
            tape[p] = (inp, a0, wt, pooled, h_pre, g, a2, H, W, Ho, Wo)
            H, W = Ho, Wo
        M = B * H * W
        raw = gemm(x, P['backbone._conv_head.weight'].view(arch.HEAD_C, -1), packed=pk)
        a = self._bn_f(tape, 'backbone._bn1', raw, M, arch.HEAD_C, 1)
        feat = rows_mean(a, B, H * W, arch.HEAD_C)
        pose = fc_small_forward(feat, P['pose_fc.weight'], P['pose_fc.bias'])
        tape['head'] = (x, feat, B, H, W)
        nbt = [v for k, v in self.buf.items() if k.endswith('num_batches_tracked')]
        if nbt:
            torch._foreach_add_(nbt, 1)      # every BatchNorm saw one more batch: one launch for all 79 counters
        return pose, tape

    def backward(self, tape, dpose, region_done=None):
        """region_done(name): called when every parameter gradient of a region has been ENQUEUED -- 'head' (conv head, bn1, pose_fc), then block
        25 .. 0, then 'stem' -- i.e. when a growing suffix of the flat gradient (named_parameters() order) is final: FlatAdam launches the bucketed
        gradient all-reduce from here, beside the rest of the backward."""
        P = self.P
        grads = {}
        x_head, feat, B, H, W = tape['head']
        D = self._dst
        staged = self.stage is not None
        pk = self.packed            # packed by this step's forward; the weights do not change between a forward and its backward
        dfeat, grads['pose_fc.weight'], grads['pose_fc.bias'] = fc_small_backward(
            dpose, feat, P['pose_fc.weight'], out=(D('pose_fc.weight'), D('pose_fc.bias')) if staged else None)
        da = rows_broadcast(dfeat, 1.0 / (H * W), B, H * W, arch.HEAD_C)
        draw = self._bn_b(tape, grads, 'backbone._bn1', da)
        wh = P['backbone._conv_head.weight']
        grads['backbone._conv_head.weight'] = wgrad(draw, x_head, out=D('backbone._conv_head.weight')).view_as(wh)
        dx = gemm(draw, wh.view(arch.HEAD_C, -1), w_is_kn=True, packed=pk)
        if region_done is not None:
            region_done('head')
        for i in reversed(range(len(arch.B3_BLOCKS))):
            k, s, e, cin, cout = arch.B3_BLOCKS[i]
            p = f'backbone._blocks.{i}.'
            inp, a0, wt, pooled, h_pre, g, a2, H, W, Ho, Wo = tape[p]
            cmid, HWo = cin * e, Ho * Wo
            skip = s == 1 and cin == cout
            dout = dx
            draw = self._bn_b(tape, grads, p + '_bn2', dout)
            wp = P[p + '_project_conv.weight']
            grads[p + '_project_conv.weight'] = wgrad(draw, a2, out=D(p + '_project_conv.weight')).view_as(wp)
            da2 = gemm(draw, wp.view(cout, cmid), w_is_kn=True, packed=pk)
            # squeeze-excite backward
            raw1, mean1, rstd1 = tape[p + '_bn1'][:3]
            dg = rows_dot_bn(da2, raw1, mean1, rstd1, P[p + '_bn1.weight'], P[p + '_bn1.bias'], B, HWo, cmid)      # sum_hw da2 * a1, a1 recomputed
            w2, w1 = P[p + '_se_expand.weight'], P[p + '_se_reduce.weight']
            dpooled, dw1, db1, dw2, db2 = se_backward(dg, g, h_pre, pooled, w1, w2, out=(D(p + '_se_reduce.weight'), D(p + '_se_reduce.bias'),
                                                      D(p + '_se_expand.weight'), D(p + '_se_expand.bias')) if staged else None)
            grads[p + '_se_reduce.weight'], grads[p + '_se_reduce.bias'] = dw1.view_as(w1), db1
            grads[p + '_se_expand.weight'], grads[p + '_se_expand.bias'] = dw2.view_as(w2), db2
            # da1 = da2 * g + dpooled / HW (gradient through the gate multiply and the pooled mean) is formed inside BatchNorm 1's backward kernels
            draw = self._bn_b(tape, grads, p + '_bn1', da2, cgate=g, cadd=dpooled, cadd_scale=1.0 / HWo, HWg=HWo)
            da0, grads[p + '_depthwise_conv.weight'] = dw_backward(a0, draw, wt, B, H, W, cmid, k, s, add=dout if (e == 1 and skip) else None,
                                                                   out=D(p + '_depthwise_conv.weight'))
            if e != 1:
                draw = self._bn_b(tape, grads, p + '_bn0', da0)
                we = P[p + '_expand_conv.weight']
                grads[p + '_expand_conv.weight'] = wgrad(draw, inp, out=D(p + '_expand_conv.weight')).view_as(we)
                # the skip connection's gradient rides on the GEMM (C = dout + draw W) instead of a separate add
                dx = gemm(draw, we.view(cmid, cin), w_is_kn=True, add=dout if skip else None, packed=pk)
            else:
                dx = da0             # (+ dout for a skip block: already added inside dw_backward)
            if region_done is not None:
                region_done(i)
        draw = self._bn_b(tape, grads, 'backbone._bn0', dx)
        cols = tape['stem']
        gstem = wgrad(draw, cols)[:, :54].reshape(arch.STEM_C, 3, 3, 6).permute(0, 3, 1, 2)
        if staged:
            D('backbone._conv_stem.weight').copy_(gstem)
        else:
            grads['backbone._conv_stem.weight'] = gstem.contiguous()
        if region_done is not None:
            region_done('stem')
        return grads


class _BackboneTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x8, drop, buffers, names, direct, *params):
        net = _Net({n: p.detach() for n, p in zip(names, params)}, buffers, stage=direct.stage if direct is not None else None)
        pose, tape = net.forward(x8.detach(), drop)
        ctx.net, ctx.tape, ctx.names, ctx.direct = net, tape, names, direct
        return pose

    @staticmethod
    def backward(ctx, dpose):
        overlap = ctx.direct is not None and ctx.direct.begin_overlapped_allreduce()
        grads = ctx.net.backward(ctx.tape, dpose.contiguous().float(), region_done=ctx.direct.region_done if overlap else None)
        ctx.tape = None
        if overlap:
            ctx.direct.finish_overlapped_allreduce()      # waits for the buckets, divides by the world size; accumulate_staged below adds the AVERAGED gradients
        if ctx.direct is not None:
            # every kernel wrote its parameter gradient straight into FlatAdam's staging buffer: ONE add accumulates all 340 of them
            # into the flat gradient (instead of 340 AccumulateGrad launches), and autograd gets nothing to accumulate
            ctx.direct.accumulate_staged()
            return (None,) * (5 + len(ctx.names))
        return (None, None, None, None, None) + tuple(grads[n] for n in ctx.names)


def _named_tensors(model):
    """(reference parameter names, the model's parameters in that order, its named buffers), from the cached tables of _modwatch
    (named_parameters() + named_buffers() walk ~300 submodules: ~1 ms of host time per step during which the GPU has nothing queued)."""
    named, buffers = _modwatch.named_tensors(model)
    cached = model.__dict__.get('_cosy_train_order')
    if cached is not None and cached[0] is named:
        return cached[1], cached[2], buffers
    names = param_names()
    missing = [n for n in names if n not in named]
    if missing or len(named) != len(names):
        raise ValueError(f'model parameters do not match the efficientnet-b3 pose network ({len(named)} vs {len(names)}; missing {missing[:3]})')
    params = [named[n] for n in names]
    model.__dict__['_cosy_train_order'] = (named, names, params)
    return names, params, buffers


def backbone_train(model, x8, drop=None):
    """pose outputs (B,9) of `model` (a PosePredictor) in train mode for the packed NHWC8 fp32 input `x8`, attached to
    the autograd graph of the model's parameters.  BatchNorm running statistics are updated in place."""
    require_device(x8)
    names, params, buffers = _named_tensors(model)
    for p_ in params:
        if p_.dtype != torch.float32 or not p_.is_contiguous():
            raise ValueError('training runs on contiguous fp32 parameters')
    direct = getattr(model, '_cosy_flat_adam', None)
    if direct is not None and not (direct.direct_grads and direct.owns(params)):
        direct = None
    return _BackboneTrainFn.apply(x8, drop or {}, buffers, names, direct, *params)


# ---------------------------------------------------------------------------------------------
# loss with its hand-written gradient
# ---------------------------------------------------------------------------------------------
class _DisentangledLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, refiner_outputs, gt, TCO_input, K_crop, points):
        B, S, Pn = gt.shape[0], gt.shape[1], points.shape[1]
        out9 = refiner_outputs.detach().float().contiguous()
        loss = torch.empty(B, device=out9.device)
        check(lib().cosy_loss_refiner_disentangled(ptr(gt), ptr(TCO_input), ptr(out9), ptr(K_crop), ptr(points), None, B, S, Pn, ptr(loss),
                                                   stream()))
        ctx.saved = (out9, gt, TCO_input, K_crop, points)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        out9, gt, TCO_input, K_crop, points = ctx.saved
        B, S, Pn = gt.shape[0], gt.shape[1], points.shape[1]
        d = torch.empty(B, 9, device=out9.device)
        dl = dloss.contiguous().float()       # held in a name: a temporary inside ptr() could be recycled before the launch
        check(lib().cosy_loss_refiner_disentangled_backward(ptr(gt), ptr(TCO_input), ptr(out9), ptr(K_crop), ptr(points), None, B, S, Pn,
                                                            ptr(dl), ptr(d), stream()))
        return d, None, None, None, None


def loss_refiner_CO_disentangled(TCO_possible_gt, TCO_input, refiner_outputs, K_crop, points):
    """(B,) loss of cosypose/lib3d/cosypose_ops.py:49-82, differentiable wrt refiner_outputs."""
    bsz = TCO_possible_gt.shape[0]
    assert TCO_input.shape == (bsz, 4, 4) and refiner_outputs.shape == (bsz, 9) and K_crop.shape == (bsz, 3, 3)
    assert points.dim() == 3 and points.shape[0] == bsz and points.shape[-1] == 3
    assert TCO_possible_gt.dim() == 4 and TCO_possible_gt.shape[-2:] == (4, 4)
    require_device(TCO_possible_gt, TCO_input, refiner_outputs, K_crop, points)
    f = lambda t: t.detach().float().contiguous()
    return _DisentangledLossFn.apply(refiner_outputs, f(TCO_possible_gt), f(TCO_input), f(K_crop), f(points))


# ---------------------------------------------------------------------------------------------
# optimizer on flat buffers, gradient averaging across ranks
# ---------------------------------------------------------------------------------------------
def _bump_version(t):
    inc = getattr(torch.autograd.graph, 'increment_version', None)
    if inc is not None:
        inc(t)
    else:
        t.add_(0)


class FlatAdam:
    """clip_grad_norm_(max_norm) + torch.optim.Adam semantics (train_pose.py:325-329, :282) as two kernel launches over
    ONE flat fp32 buffer.  Construction re-homes the parameters (and their .grad) as views of flat buffers, in
    named_parameters() order; the module keeps working as before."""

    def __init__(self, model, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_grad_norm=0.5, direct_grads=True, overlap_allreduce=False,
                 bucket_bytes=8 << 20):
        params = [p for p in model.parameters()]
        dev = params[0].device
        require_device(params[0])
        n = sum(p.numel() for p in params)
        self.flat = torch.empty(n, device=dev)
        self.grad = torch.zeros(n, device=dev)
        off = 0
        for p in params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p.data)
            p.grad = self.grad[off:off + k].view_as(p.data)
            off += k
        self.params = params
        # direct_grads: the backward of backbone_train writes every parameter gradient into `gstage` (same layout as `grad`) and adds the
        # whole buffer to `grad` with one launch; autograd's 340 per-parameter AccumulateGrad launches (~1.5 ms per step) disappear.
        # Semantics are those of .grad accumulation (several backwards before a step add up).  NOT with a DistributedDataParallel
        # wrapper -- its reducer hangs on the per-parameter hooks that never fire then; average with allreduce_gradients(opt) instead
        # (or pass direct_grads=False).
        self.direct_grads = bool(direct_grads)
        # overlap_allreduce (default False: a backward issues NO collective unless asked -- a process group that exists for sharded inference, a
        # rank-local gradient check or an uneven number of backwards per rank would otherwise hang on an implicit one; train_loop / bench_train.py
        # turn it on for their data-parallel steps; None = automatic: on when a group with more than one rank exists): the backward launches the gradient
        # all-reduce itself, bucket by bucket (>= bucket_bytes of finished gradients each: head + late blocks first), beside the rest of the
        # backward; the staged gradients are averaged before they are accumulated, `reduced` tells train_loop / allreduce_gradients that this
        # backward's gradients are already averaged.  Buckets are slices of ONE buffer: elementwise sums, so the result equals the single
        # all-reduce bit for bit at 2 ranks (any order of two addends) and up to the collective's own chunking beyond.
        self.overlap_allreduce = overlap_allreduce
        self.bucket_bytes = int(bucket_bytes)
        self.n_backward = self.n_reduced = 0      # direct-path backwards since zero_grad, and how many of them averaged their own gradients
        self._pending, self._hi = [], n
        self.gstage = torch.zeros(n, device=dev)
        self.stage, off = {}, 0
        self._region_lo = {}          # region ('stem', block index, 'head') -> offset of its first parameter in the flat layout
        for name, p in model.named_parameters():
            self.stage[name] = self.gstage[off:off + p.numel()].view_as(p.data)
            m_ = name.split('.')
            region = int(m_[2]) if name.startswith('backbone._blocks.') else ('stem' if name in ('backbone._conv_stem.weight', 'backbone._bn0.weight', 'backbone._bn0.bias') else 'head')
            self._region_lo.setdefault(region, off)
            off += p.numel()
        # the regions must tile the buffer in the order the backward finishes them backwards: stem < block 0 < ... < block 25 < head
        order = ['stem'] + [i for i in range(len(arch.B3_BLOCKS))] + ['head']
        los = [self._region_lo.get(r, -1) for r in order]
        self._regions_ok = all(v >= 0 for v in los) and los == sorted(los) and los[0] == 0
        model.__dict__['_cosy_flat_adam'] = self          # found by backbone_train (not a submodule / parameter: invisible to state_dict)
        self.m = torch.zeros(n, device=dev); self.v = torch.zeros(n, device=dev)
        self.norm_coef = torch.ones(2, device=dev)
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, clip_grad_norm
        self.step_count = 0

    def begin_overlapped_allreduce(self):
        """-> True when this backward is to launch its own bucketed all-reduce (called by the backward node before it starts)"""
        import torch.distributed as dist
        on = self.overlap_allreduce
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if on is None:
            on = multi
        if not (on and multi and self._regions_ok):
            return False
        self._pending, self._hi = [], self.gstage.numel()
        return True

    def region_done(self, region):
        """the gradients of `region` and of everything behind it in the flat layout have been enqueued: all-reduce the finished suffix once it holds
        bucket_bytes (and whatever is left when the stem is done)"""
        import torch.distributed as dist
        lo = self._region_lo[region]
        if (self._hi - lo) * 4 >= self.bucket_bytes or region == 'stem':
            if self._hi > lo:
                self._pending.append(dist.all_reduce(self.gstage[lo:self._hi], op=dist.ReduceOp.SUM, async_op=True))
            self._hi = lo

    def finish_overlapped_allreduce(self):
        import torch.distributed as dist
        assert self._hi == 0, 'the backward did not report every region'
        for work in self._pending:
            work.wait()
        self._pending = []
        self.gstage.div_(dist.get_world_size())
        self.n_reduced += 1

    def owns(self, params):
        """the given parameters are exactly this optimizer's, still living in its flat buffer"""
        if len(params) != len(self.params):
            return False
        base, off = self.flat.data_ptr(), 0
        for p, q in zip(params, self.params):
            if p is not q or p.data_ptr() != base + 4 * off:
                return False
            off += p.numel()
        return True

    def accumulate_staged(self):
        """grad += gstage (one launch).  Parameters whose .grad was dropped (zero_grad(set_to_none=True)) restart from zero and ones
        that were given a foreign .grad tensor are copied in first, exactly as step() treats them."""
        self._collect_stray_gradients()
        self.grad.add_(self.gstage)
        self.n_backward += 1

    @property
    def reduced(self):
        """every backward accumulated since zero_grad averaged its own gradients over the ranks (overlap_allreduce)"""
        return self.n_backward > 0 and self.n_reduced == self.n_backward

    def zero_grad(self):
        self.grad.zero_()
        self.n_backward = self.n_reduced = 0
        off = 0
        for p in self.params:       # re-attach views in case something set .grad to None
            if p.grad is None:
                p.grad = self.grad[off:off + p.numel()].view_as(p.data)
            off += p.numel()

    def _collect_stray_gradients(self):
        """`model.zero_grad()` / a torch optimizer's `zero_grad(set_to_none=True)` detach p.grad from the flat buffer: the
        next backward then allocates fresh .grad tensors and the flat buffer would silently stay zero.  Copy such
        gradients in and re-attach the views (a missing .grad counts as zero, as in torch.optim)."""
        off = 0
        base = self.grad.data_ptr()
        for p in self.params:
            k = p.numel()
            g = p.grad
            if g is None or g.data_ptr() != base + 4 * off:
                view = self.grad[off:off + k].view_as(p.data)
                if g is None:
                    view.zero_()
                else:
                    view.copy_(g)
                p.grad = view
            off += k

    def step(self):
        """-> total gradient norm before clipping (device scalar)"""
        self._collect_stray_gradients()
        n = self.flat.numel()
        check(lib().cosy_grad_norm_clip(ptr(self.grad), n, float(self.max_norm or 0.0), ptr(self.norm_coef), ptr(_workspace(self.flat.device)),
                                        stream()))
        self.step_count += 1
        check(lib().cosy_adam_step(ptr(self.flat), ptr(self.grad), ptr(self.m), ptr(self.v), n, self.lr, self.betas[0], self.betas[1],
                                   self.eps, self.wd, self.step_count, ptr(self.norm_coef), stream()))
        _bump_version(self.flat)     # the kernel wrote through raw pointers: let torch (and the inference engine's cache) know
        return self.norm_coef[0]


def allreduce_gradients(flat_grad, force=False):
    """DDP's gradient averaging as ONE all-reduce of the flat gradient buffer (42.8 MB fp32 for this network).
    `flat_grad`: the buffer, or a FlatAdam (gradients that were detached from its buffer are collected first, so that
    what is reduced is what the step will use).  force=True runs the collective in a 1-rank group too (RCCL smoke test)."""
    import torch.distributed as dist
    if isinstance(flat_grad, FlatAdam):
        if flat_grad.reduced and not force:       # EVERY backward since zero_grad launched its own bucketed all-reduce (FlatAdam.overlap_allreduce): already averaged
            return flat_grad.grad
        if flat_grad.n_reduced and not force:
            raise RuntimeError(f'allreduce_gradients: {flat_grad.n_reduced} of the {flat_grad.n_backward} backwards accumulated since zero_grad averaged their own '
                               'gradients (overlap_allreduce) and the others did not: the buffer mixes averaged and local gradients')
        flat_grad._collect_stray_gradients()
        flat_grad = flat_grad.grad
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        flat_grad.div_(dist.get_world_size())
    return flat_grad
