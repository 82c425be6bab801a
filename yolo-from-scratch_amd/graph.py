"""Static network plans over NHWC buffers: the host-side scheduler of the HIP kernels.

A module tree is *traced once* per (input shape, mode) into a list of layer records over symbolic
NHWC buffers; the records are lowered to two flat op lists (forward, backward) of `yh_op` structs
holding raw device pointers.  Running a step is then two C calls (`yh_run`) -- no per-layer Python,
no autograd graph, no allocation -- which is what makes the lists hipGraph-friendly.

Design points (all about HBM traffic, the bound for this small-channel network):
  * torch.cat never materialises: producers write straight into channel slices of a shared buffer
    (a view = buffer + channel offset, ld = buffer channels);
  * nearest-x2 upsampling is folded into the producer's BN+SiLU write, the residual add into the
    same pass; the head outputs are already (B,G,G,3,5+nc) because the layout is NHWC;
  * BatchNorm statistics come out of the conv epilogue; dY overwrites Y in place in the backward.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L

BN_EPS_DEFAULT = 1e-5
# bumped whenever this package's kernels modify parameters or BatchNorm buffers in place (see Plan.weights_state)
WEIGHTS_EPOCH = [0]
# bumped whenever this package moves parameter / buffer STORAGE (HipTrainer adopting the parameters into its flat buffer,
# Module._apply: .to() / .cuda() / .float(), load_state_dict): plans and captured hipGraphs hold raw addresses
PARAM_GENERATION = [0]


def _on_reregistration(module, name, value):
    """Global torch hook: a parameter or buffer attribute that already held a tensor is being REPLACED (`conv.weight =
    nn.Parameter(...)`, `bn.running_mean = t`, load_state_dict(assign=True) through any parent module).  Plans and captured
    hipGraphs hold the old tensor's address: bump the generation so every holder re-validates (the per-image check of
    InferenceSession compares this counter instead of walking the module tree)."""
    for store in ("_parameters", "_buffers"):
        d = module.__dict__.get(store)
        if d is not None and d.get(name) is not None and d.get(name) is not value:
            PARAM_GENERATION[0] += 1
            return None
    return None


torch.nn.modules.module.register_module_parameter_registration_hook(_on_reregistration)
torch.nn.modules.module.register_module_buffer_registration_hook(_on_reregistration)


def invalidate_folded_weights():
    """Tell every eval plan / InferenceSession that parameter or BatchNorm-buffer VALUES changed behind torch's version
    counters.  The package calls this itself wherever its own code writes parameter storage (optimizer kernels, training
    forwards, HipTrainer's broadcast, load_state_dict); user code must call it after writes torch cannot see on the
    registered tensors: `p.data.mul_()` / `p.data.copy_()` (`.data` carries its own version counter -- the usual EMA idiom),
    writes through an aliasing view such as `trainer.flat_p`, and in-place collectives on parameters."""
    WEIGHTS_EPOCH[0] += 1


def state_token():
    """One cheap comparison for 'nothing this package knows of has touched the weights or their addresses'."""
    return (WEIGHTS_EPOCH[0], PARAM_GENERATION[0])


def _rup4(c: int) -> int:
    return (c + 3) // 4 * 4


def _rup8(c: int) -> int:
    return (c + 7) // 8 * 8


class Buffer:
    """An NHWC activation tensor (B,H,W,C) -- fp32, or bf16 on the bf16 path -- plus, lazily, its gradient tensor.
    The gradient may have its own type and pixel stride: the bf16 path's head outputs stay fp32 (they feed the fp32
    loss) while their gradient is bf16 with the channel axis zero-padded to a multiple of 8 (grad_C)."""

    def __init__(self, plan: "Plan", B: int, H: int, W: int, C: int, name: str, dtype: Optional[torch.dtype] = None,
                 grad_C: Optional[int] = None):
        self.plan, self.B, self.H, self.W, self.C, self.name = plan, B, H, W, C, name
        self.dtype = dtype if dtype is not None else plan.act_dtype
        self.grad_dtype, self.grad_C = plan.act_dtype, (grad_C if grad_C is not None else C)
        self.data = torch.empty(B, H, W, C, device=plan.device, dtype=self.dtype)
        self._grad: Optional[torch.Tensor] = None
        self.grad_cover: List[Tuple[int, int]] = []   # channel ranges already written in this backward
        self._icoef: Optional[torch.Tensor] = None

    @property
    def icoef(self) -> torch.Tensor:
        """Input-prologue table of the buffer's channels, rows [scale | shift | gate] (see yh_prologue in csrc/common.h): channels
        whose producer's BatchNorm + SiLU is applied by the READERS (the activation is never materialised, the buffer holds the raw
        convolution output) carry the producer's scale / shift (rewritten by its bn_finalize every step) and gate 1; all others
        the identity (1, 0, gate 0)."""
        if self._icoef is None:
            ld = _rup4(self.C) + 4
            t = torch.zeros(3, ld, device=self.plan.device, dtype=torch.float32)
            t[0].fill_(1.0)
            self._icoef = t
        return self._icoef

    @property
    def grad(self) -> torch.Tensor:
        if self._grad is None:      # zeros: padding channels of a padded gradient must stay finite (they meet zero weights)
            self._grad = torch.zeros(self.B, self.H, self.W, self.grad_C, device=self.plan.device, dtype=self.grad_dtype)
        return self._grad

    def view(self, off: int = 0, C: Optional[int] = None) -> "View":
        return View(self, off, self.C - off if C is None else C)


@dataclass
class View:
    buf: Buffer
    off: int
    C: int

    @property
    def B(self): return self.buf.B
    @property
    def H(self): return self.buf.H
    @property
    def W(self): return self.buf.W
    @property
    def ld(self): return self.buf.C
    @property
    def ldg(self): return self.buf.grad_C
    @property
    def M(self): return self.buf.B * self.buf.H * self.buf.W
    def ptr(self) -> int: return self.buf.data.data_ptr() + self.buf.data.element_size() * self.off
    def gptr(self) -> int: return self.buf.grad.data_ptr() + self.buf.grad.element_size() * self.off
    def icoef_ptr(self, row: int = 0) -> int: return self.buf.icoef.data_ptr() + 4 * (row * self.buf.icoef.shape[1] + self.off)
    @property
    def icoef_ld(self) -> int: return self.buf.icoef.shape[1]


def _same_view(a: "View", b: "View") -> bool:
    return a.buf is b.buf and a.off == b.off and a.C == b.C


@dataclass
class ConvRec:
    """conv (+bias) [+ BatchNorm + SiLU (+residual) (+x2 upsample on write)]"""
    x: View
    out: View
    weight: torch.nn.Parameter
    bias: Optional[torch.nn.Parameter]
    bn: Optional[torch.nn.BatchNorm2d]
    k: int
    s: int
    residual: Optional[View] = None
    upsample: bool = False
    # filled by the planner
    cin: int = 0            # padded input channels seen by the kernels
    cout: int = 0
    Ho: int = 0
    Wo: int = 0
    y: Optional[torch.Tensor] = None        # raw conv output (B,Ho,Wo,cout); dY overwrites it in the backward
    coef: Optional[torch.Tensor] = None
    part: Optional[torch.Tensor] = None
    wf: Optional[torch.Tensor] = None
    wb: Optional[torch.Tensor] = None
    ldwf: int = 0
    ldwb: int = 0
    need_dx: bool = True
    lane: int = 0
    wino_f: bool = False    # forward on the Winograd F(2x2,3x3) kernel
    wino_b: bool = False    # backward-data on the Winograd kernel
    wino_w: bool = False    # backward-weight in the Winograd domain
    pw_w: bool = False      # backward-weight on the pointwise (1x1) kernel
    pw_f: bool = False      # forward on the pointwise GEMM kernel
    s2m_b: bool = False     # stride-2 backward-data with the column parities merged into the channel axis
    narrow_f: bool = False  # forward on the direct kernel for the narrow high-resolution 3x3 layers
    narrow_b: bool = False  # stride-1 backward-data on the same kernel (flipped taps)
    narrow_w: bool = False  # weight gradient on the direct pixel-reduction kernel
    s2l_f: bool = False     # forward on the LDS-staged 3x3 stride-2 kernel (conv_s2.hip)
    cin_k: int = 0          # bf16 plan: input channels the narrow kernels read (4 of the first layer's 8 padded ones)
    fwd2: bool = False      # forward fused with the sibling pointwise conv (one GEMM, N = cout1 + cout2)
    nblk: int = 0           # BatchNorm partial-sum rows written by the forward kernel
    pw_b: bool = False      # backward-data on the pointwise GEMM kernel
    pair: Optional["ConvRec"] = None       # sibling pointwise conv reading the same input (fused backward-data)
    pair_first: bool = False
    conv: Optional[torch.nn.Conv2d] = None  # the parameter container (staleness checks look the parameters up again)
    virtual: bool = False   # the activation is never materialised: the conv writes its raw output into `out`, readers apply BN + SiLU
    x_fused: bool = False   # the input view has virtual channels: forward / weight-gradient kernels run the input prologue
    yp: int = 0             # address and pixel stride of the raw conv output (private tensor `y`, or the `out` view when virtual)
    ldy: int = 0
    dyp: int = 0            # where the BatchNorm backward writes dY: over Y in place, or (virtual) into the private tensor `y` -- the
    lddy: int = 0           # raw output in the shared buffer is still being read by the consumers' weight gradients on the side lane
    kcout_b: int = 0        # K rows of the backward-data pack (> cout: a head whose gradient rows are zero-padded; 0: cout)


@dataclass
class PoolRec:
    x: View
    out: View
    arg: Optional[torch.Tensor] = None
    lane: int = 0
    fused: bool = False     # eval: computed by the preceding pool's three-stage launch


@dataclass
class SyncRec:
    """fork (side lane may start) / join (main lane waits for the side lane) marker in the forward order"""
    kind: str


class Plan:
    """Trace target + compiled op lists for one module at one input shape / mode."""

    def __init__(self, device: torch.device, in_shape: Tuple[int, int, int, int], training: bool, need_input_grad: bool,
                 dtype: str = "f32"):
        if dtype not in ("f32", "bf16"):
            raise ValueError(f"compute dtype must be 'f32' or 'bf16', got {dtype!r}")
        if dtype == "bf16" and not training:
            raise NotImplementedError("the bf16 path covers training plans; eval plans run the fp32 fused kernels")
        self.device, self.training, self.need_input_grad = device, training, need_input_grad
        self.dtype = dtype
        self.bf16 = dtype == "bf16"
        self.act_dtype = torch.bfloat16 if self.bf16 else torch.float32
        self.B, self.Cimg, self.Himg, self.Wimg = in_shape
        self.recs: List[object] = []
        self.buffers: List[Buffer] = []
        self.outputs: List[Tuple[View, str]] = []       # (view, "nhwc_heads" | "nchw")
        self.input = self.new_buffer(self.B, self.Himg, self.Wimg, (_rup8 if self.bf16 else _rup4)(self.Cimg), "input")
        self.generation = 0
        self.lane = 0                      # lane given to records traced from now on (see side_lane())
        self.fwd_ops = self.bwd_ops = None
        self.param_ptrs: List[int] = []
        self._sig_holders = None
        self._grad_of = None
        self.prep_ops = None               # eval plans: BatchNorm folding, run once per weight state
        self._fold_tensors = None
        self._folded_state = None

    # ---- tracing API used by the modules ------------------------------------------------------
    def new_buffer(self, B, H, W, C, name="act", dtype=None, grad_C=None) -> Buffer:
        b = Buffer(self, B, H, W, C, name, dtype, grad_C)
        self.buffers.append(b)
        return b

    def conv(self, x: View, conv: torch.nn.Conv2d, bn: Optional[torch.nn.BatchNorm2d], out: Optional[View] = None,
             residual: Optional[View] = None, upsample: bool = False) -> View:
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        if conv.kernel_size[0] != conv.kernel_size[1] or k not in (1, 3) or s not in (1, 2) or p != k // 2 \
                or conv.stride[0] != conv.stride[1] or conv.groups != 1 or conv.dilation != (1, 1):
            raise NotImplementedError(f"HIP conv path supports square k in (1,3), stride in (1,2), padding k//2; got {conv}")
        if (k, s) == (1, 2):
            raise NotImplementedError("1x1 stride-2 convolutions are not part of this network")
        if conv.in_channels != x.C and (_rup8 if self.bf16 else _rup4)(conv.in_channels) != x.C:
            raise ValueError(f"conv expects {conv.in_channels} input channels, got a view with {x.C}")
        if self.bf16 and (x.C % 8 or x.ld % 8 or x.off % 8 or (bn is not None and conv.out_channels % 8)):
            raise NotImplementedError("the bf16 path needs channel counts, views and strides in multiples of 8")
        Ho, Wo = (x.H + 2 * p - k) // s + 1, (x.W + 2 * p - k) // s + 1
        cout = conv.out_channels
        if bn is not None and cout % 4:
            raise NotImplementedError("Conv+BN+SiLU on the HIP path needs out_channels % 4 == 0")
        f = 2 if upsample else 1
        if out is None and self.bf16 and bn is None:
            # a plain conv output on the bf16 path (the detection heads): fp32 values for the fp32 loss, bf16 gradient
            # with the channel axis zero-padded to a multiple of 8 (the backward GEMMs read 16-byte pieces)
            out = self.new_buffer(x.B, Ho * f, Wo * f, cout, "head", dtype=torch.float32, grad_C=_rup8(cout)).view()
        if out is None and bn is None:
            # a plain conv output (the detection heads, 3 (5 + nc) = 18 / 255 channels): dense values for the loss, but a gradient whose
            # pixel rows are zero-padded to a multiple of 4 channels -- the backward GEMMs and the bias column sum then read 16-byte
            # pieces (at nc = 80 the 255-float rows ran every one of them on its scalar-load variant)
            out = self.new_buffer(x.B, Ho * f, Wo * f, cout, "head", grad_C=_rup4(cout)).view()
        if out is None:
            out = self.new_buffer(x.B, Ho * f, Wo * f, _rup4(cout)).view()
        if self.bf16 and bn is not None and (out.off % 8 or out.ld % 8):
            raise NotImplementedError("the bf16 path needs output views aligned to 8 channels")
        if self.bf16 and bn is None and out.buf.dtype != torch.float32:
            raise NotImplementedError("plain conv outputs on the bf16 path are fp32 head tensors")
        if (out.H, out.W, out.C) != (Ho * f, Wo * f, cout):
            raise ValueError(f"output view {(out.H, out.W, out.C)} does not match conv result {(Ho * f, Wo * f, cout)}")
        if residual is not None and (residual.H, residual.W, residual.C) != (Ho, Wo, cout):
            raise ValueError("residual shape mismatch")
        self.recs.append(ConvRec(x, out, conv.weight, conv.bias, bn, k, s, residual, upsample, cin=x.C, cout=cout,
                                 Ho=Ho, Wo=Wo, lane=self.lane, conv=conv))
        return out

    # independent sub-graphs: `with plan.side_lane():` traces the enclosed layers onto the side stream;
    # a fork marker is emitted on entry (the side lane sees everything traced before), the join is emitted
    # by `join()` (or implicitly at the end of the op list)
    def side_lane(self):
        plan = self

        enabled = True

        class _Ctx:
            def __enter__(self_inner):
                if enabled:
                    plan.recs.append(SyncRec("fork"))
                    plan.lane = 1

            def __exit__(self_inner, *exc):
                plan.lane = 0
        return _Ctx()

    def join(self):
        self.recs.append(SyncRec("join"))

    def pool5(self, x: View, out: View) -> View:
        if x.C % 4 or (x.H, x.W, x.C) != (out.H, out.W, out.C):
            raise ValueError("pool5: shape mismatch or channels not a multiple of 4")
        self.recs.append(PoolRec(x, out, lane=self.lane))
        return out

    def mark_output(self, v: View, kind: str):
        self.outputs.append((v, kind))

    # ---- lowering -----------------------------------------------------------------------------
    def params(self) -> List[torch.nn.Parameter]:
        seen, out = set(), []
        for r in self.recs:
            if isinstance(r, ConvRec):
                for p in (r.weight, r.bias, r.bn.weight if r.bn is not None else None,
                          r.bn.bias if r.bn is not None else None):
                    if p is not None and id(p) not in seen:
                        seen.add(id(p))
                        out.append(p)
        return out

    # ---- kernel-family routing (fp32 training plans) ---------------------------------------------------------------------
    def _route_f32(self, r: "ConvRec", special: bool):
        """Which kernel family runs each pass of one convolution.  Decided for every record BEFORE anything is lowered: the fusion
        planner (_plan_fusion) needs to know whether every reader of a tensor can apply the producer's BatchNorm + SiLU itself."""
        lib = L.lib()
        use_wino = use_pw = use_pwg = use_s2m = use_narrow = special
        r.ldwb = _rup4(r.cin)
        r.ldwf = _rup4(r.cout)
        r.need_dx = self.training and (self.need_input_grad or r.x.buf is not self.input)
        wino_ok = use_wino and r.k == 3 and r.s == 1 and r.cin == r.weight.shape[1] and r.x.H % 2 == 0 and r.x.W % 2 == 0 and r.x.ld % 4 == 0
        r.wino_f = wino_ok and r.cin % 16 == 0 and r.cin <= 2048
        r.wino_b = wino_ok and r.need_dx and r.cout % 16 == 0 and r.cout <= 2048
        r.wino_w = wino_ok and r.cin % 32 == 0 and r.cout % 32 == 0 and r.x.W >= 4
        r.pw_w = use_pw and r.k == 1 and r.s == 1 and r.cin == r.weight.shape[1]
        # pointwise GEMM kernels: measured faster than the gather-GEMM except when both K and N are >= 256
        pw_ok = use_pwg and r.k == 1 and r.s == 1 and r.cin == r.weight.shape[1] and r.x.ld % 4 == 0
        r.pw_f = pw_ok and r.cin % 8 == 0 and not (r.cin >= 256 and r.cout >= 256)
        r.s2m_b = use_s2m and r.k == 3 and r.s == 2 and r.need_dx and r.cin <= 16 and r.x.ld == r.cin and r.x.W % 2 == 0 \
            and r.cin == r.weight.shape[1]
        # narrow high-resolution 3x3 layers: direct kernel (LDS halo patch, filter in registers, 16-wide MFMA tiles)
        nar_ok = use_narrow and r.k == 3 and r.cin == r.weight.shape[1] and r.x.ld % 4 == 0 and r.x.off % 4 == 0
        r.narrow_f = bool(nar_ok and lib.yh_conv_narrow_ok(r.cin, r.cout, 3, r.s))
        if use_narrow and r.k == 3 and r.s == 2 and r.cin == 4 and r.x.ld == 4 and r.cout == 16 and r.ldwf == 16 \
                and lib.yh_conv_narrow_ok(r.cin, r.cout, 3, r.s):
            r.narrow_f = True                     # first layer: the same direct MFMA kernel with CIN = 4 (padded) channels
        r.s2l_f = bool(special and not r.narrow_f and r.k == 3 and r.s == 2 and r.cin == r.weight.shape[1] and r.x.ld % 4 == 0 and
                       r.x.off % 4 == 0 and lib.yh_conv_s2_ok(r.x.B, r.x.H, r.x.W, r.cin, r.cout))
        r.narrow_b = bool(nar_ok and r.s == 1 and r.need_dx and lib.yh_conv_narrow_ok(r.cout, r.cin, 3, 1))
        if r.narrow_f:
            r.wino_f = False
        if r.narrow_b:
            r.wino_b = False
        if nar_ok and r.s == 2 and r.need_dx and lib.yh_conv_narrow_dgrad_s2_ok(r.cin, r.cout):
            r.narrow_b, r.s2m_b = True, False      # the stride-2 form of the direct backward-data kernel
        r.narrow_w = bool(use_narrow and r.k == 3 and r.x.ld % 4 == 0
                          and r.x.off % 4 == 0 and lib.yh_conv_narrow_bwd_weight_ok(r.cin, r.weight.shape[1], r.cout, 3, r.s))
        if r.narrow_w:
            r.wino_w = False
        if r.pair is None:
            r.pw_b = pw_ok and r.need_dx and r.cout % 8 == 0 and not (r.cout >= 256 and r.cin >= 256)
        elif r.pair_first:
            kpair = r.cout + r.pair.cout
            r.pw_b = r.pair.pw_b = pw_ok and r.need_dx and r.cout % 8 == 0 and r.pair.cout % 8 == 0 and \
                not (kpair >= 256 and r.cin >= 256)

    def _plan_fusion(self, special: bool):
        """Decide which Conv+BN+SiLU outputs are never materialised ("virtual"): the convolution writes its RAW output straight
        into the output view (a channel slice of the consumer's buffer), bn_finalize publishes scale / shift in the buffer's
        prologue table, and every reader -- forward kernel, weight-gradient kernel, residual add -- applies scale, shift and SiLU
        while it stages its operand (train.py:253-265 executed at the consumer).  Saves the whole bn_silu_fwd round trip (8 bytes
        per element fp32).  A tensor stays materialised when it carries a residual or an upsample, is a plan output, feeds a
        max-pool, or has a reader whose kernel family has no prologue."""
        readers: Dict[int, List[tuple]] = {}
        for r in self.recs:
            if isinstance(r, ConvRec):
                readers.setdefault(id(r.x.buf), []).append((r.x.off, r.x.C, "conv", r))
                if r.residual is not None:
                    readers.setdefault(id(r.residual.buf), []).append((r.residual.off, r.residual.C, "res", r))
            elif isinstance(r, PoolRec):
                readers.setdefault(id(r.x.buf), []).append((r.x.off, r.x.C, "pool", r))
        outs = [(id(v.buf), v.off, v.C) for v, _ in self.outputs]
        enabled = special and os.environ.get("YH_FUSE_ACT", "1") != "0"

        def overlap(o1, c1, o2, c2):
            return o1 < o2 + c2 and o2 < o1 + c1
        for r in self.recs:
            if not isinstance(r, ConvRec) or r.bn is None:
                continue
            v = r.out
            ok = enabled and self.training and not self.bf16 and not r.upsample and r.residual is None and v.off % 4 == 0 and v.C % 4 == 0
            ok = ok and not any(b == id(v.buf) and overlap(v.off, v.C, o, c) for b, o, c in outs)
            n_read = 0
            for off, C, kind, c in readers.get(id(v.buf), []):
                if not ok or not overlap(v.off, v.C, off, C):
                    continue
                n_read += 1
                if kind == "pool":
                    ok = False
                elif kind == "conv":
                    ok = self._fwd_prologue_ok(c) and self._wgrad_prologue_ok(c)
            r.virtual = bool(ok and n_read > 0)
        for r in self.recs:                      # readers of virtual channels
            if isinstance(r, ConvRec):
                r.x_fused = any(isinstance(q, ConvRec) and q.virtual and q.out.buf is r.x.buf and overlap(q.out.off, q.out.C, r.x.off, r.x.C)
                                for q in self.recs)
        for r in self.recs:
            if isinstance(r, ConvRec) and r.virtual:
                r.out.buf.icoef[2, r.out.off: r.out.off + r.out.C] = 1.0

    def _fwd_prologue_ok(self, c: "ConvRec") -> bool:
        """Forward kernel families that can apply the producer's BatchNorm + SiLU while staging their input."""
        lib = L.lib()
        if c.x.off % 4 or c.x.ld % 4 or c.cin % 4 or c.cin != c.weight.shape[1]:
            return False
        M = c.x.B * c.Ho * c.Wo
        if c.narrow_f:
            return c.cin == 16
        if c.wino_f or c.s2l_f:
            return True
        if c.fwd2:
            return bool(lib.yh_conv_pw_prologue_ok(M, c.cin, c.cout + c.pair.cout))
        if c.pw_f:
            return bool(lib.yh_conv_pw_prologue_ok(M, c.cin, c.cout))
        return True                                   # gather-GEMM: 16-byte staging of 4-channel pieces

    def _wgrad_prologue_ok(self, c: "ConvRec") -> bool:
        lib = L.lib()
        if c.x.off % 4 or c.x.ld % 4 or c.cin % 4 or c.cin != c.weight.shape[1]:
            return False
        if c.narrow_w:
            return c.cin == 16
        if c.wino_w or c.pw_w:
            return True
        return bool(lib.yh_conv_bwd_weight_prologue_ok(c.x.B, c.x.H, c.x.W, c.cin, c.cout, c.k, c.s))

    def _res_fused(self, r: "ConvRec") -> bool:
        v = r.residual
        return v is not None and any(isinstance(q, ConvRec) and q.virtual and q.out.buf is v.buf and
                                     q.out.off < v.off + v.C and v.off < q.out.off + q.out.C for q in self.recs)

    def compile(self, grad_of: Optional[Dict[int, torch.Tensor]] = None):
        """Allocate per-layer scratch and emit the op lists.  grad_of maps id(param) -> tensor that
        receives its gradient (views of a flat buffer); required when training."""
        lib = L.lib()
        dev = self.device
        if self.bf16:
            return self._compile_bf16(grad_of)
        f32 = dict(device=dev, dtype=torch.float32)
        fwd: List[L.YhOp] = []
        keep: List[torch.Tensor] = []          # tensors referenced only through raw pointers
        packs: List[tuple] = []                # one descriptor per conv for the single pack launch
        folds: List[tuple] = []                # inference: BN folded into the packed weights
        winos: List[tuple] = []                # Winograd weight transforms (forward and backward-data)
        latpacks: List[tuple] = []             # k-quad packs of the latency-oriented inference kernel (conv_lat.hip)
        # YH_GENERIC=1: every convolution on the generic gather-GEMM / wgrad kernels instead of the specialised families
        # (Winograd, pointwise, narrow, merged stride-2): the two are independent product paths that must agree
        # (tests/test_gpu_model.py::test_full_size_step_properties); the only planner switch besides YH_EVAL_FAST
        special = self.training and os.environ.get("YH_GENERIC", "0") != "1"
        use_wino = use_pw = use_pwg = use_s2m = use_narrow = special
        s2m_packs: List[L.YhOp] = []
        pwpacks: List[tuple] = []              # k-quad interleaved weights of the pointwise GEMM kernels
        if self.training:
            # sibling pointwise convs (C3 conv1 / conv2) share one backward-data GEMM: K = Cout1 + Cout2
            groups: Dict[tuple, List[ConvRec]] = {}
            for r in self.recs:
                if isinstance(r, ConvRec) and r.k == 1 and r.s == 1 and r.bn is not None and r.cout % 4 == 0:
                    groups.setdefault((id(r.x.buf), r.x.off, r.x.C), []).append(r)
            for grp in groups.values():
                if len(grp) == 2 and grp[0].cout == grp[1].cout and (self.need_input_grad or grp[0].x.buf is not self.input):
                    grp[0].pair, grp[1].pair = grp[1], grp[0]
                    grp[0].pair_first = True
                    a = grp[0]           # forward fusion (one GEMM, N = cout1 + cout2): decided here so the lowering below
                    a.fwd2 = grp[1].fwd2 = (  # can order the fused launch before a side-lane fork
                        use_pwg and a.cin == a.weight.shape[1] and a.x.ld % 4 == 0 and a.cin % 8 == 0
                        and not (a.cin >= 256 and a.cout >= 256))
        if self.training:
            for r in self.recs:
                if isinstance(r, ConvRec):
                    self._route_f32(r, special)
            self._plan_fusion(special)
        deferred_fork = False
        for ri, r in enumerate(self.recs):
            if isinstance(r, SyncRec):
                nxt = self.recs[ri + 1] if ri + 1 < len(self.recs) else None
                if r.kind == "fork" and isinstance(nxt, ConvRec) and nxt.fwd2 and nxt.pair_first and nxt.lane == 1:
                    deferred_fork = True     # the fused sibling conv runs on the main lane; fork right after it (below)
                else:
                    fwd.append(_op(L.OP_FORK if r.kind == "fork" else L.OP_JOIN))
                continue
            ln = r.lane
            if isinstance(r, ConvRec):
                kk = r.k * r.k
                fused_second = r.fwd2 and r.pair is not None and not r.pair_first and r.wf is not None   # shares its sibling's matrix
                if not self.training:
                    r.ldwb = _rup4(r.cin)
                if not fused_second:
                    r.ldwf = _rup4(r.cout)
                    r.wf = torch.empty(kk * r.cin * r.ldwf, **f32)
                if not self.training:
                    # eval: one fused kernel per conv -- conv + folded-BN bias + SiLU (+residual) (+x2 upsample).  BatchNorm
                    # is folded into an OIHW copy of the weights (yh_fold_oihw_multi), which then goes through the weight
                    # transform of whichever kernel family runs the layer: Winograd / pointwise GEMM for layers with enough
                    # workgroups to fill the chip (eval_epoch, predict_batch), the gather-GEMM with in-launch split-K for
                    # the small-M layers of batch-1 inference.
                    bn = r.bn
                    cin_real = r.weight.shape[1]
                    fbias = torch.empty(r.cout, **f32)
                    wfold = torch.empty(r.cout * cin_real * kk, **f32)
                    keep += [fbias, wfold]
                    folds.append((r.weight.data_ptr(), r.bias.data_ptr() if r.bias is not None else 0,
                                  bn.weight.data_ptr() if bn is not None else 0, bn.bias.data_ptr() if bn is not None else 0,
                                  bn.running_mean.data_ptr() if bn is not None else 0,
                                  bn.running_var.data_ptr() if bn is not None else 0, wfold.data_ptr(), fbias.data_ptr(),
                                  r.cout, cin_real * kk, float(bn.eps) if bn is not None else 0.0, 0))
                    fast = os.environ.get("YH_EVAL_FAST", "1") != "0"
                    M = r.x.B * r.Ho * r.Wo
                    aligned = r.cin == cin_real and r.x.ld % 4 == 0
                    wino_e = fast and aligned and r.k == 3 and r.s == 1 and r.x.H % 2 == 0 and r.x.W % 2 == 0 and r.cin % 16 == 0 \
                        and r.cin <= 2048 and ((M // 4 + 31) // 32) * ((r.cout + 63) // 64) >= 192
                    pw_e = fast and aligned and r.k == 1 and r.s == 1 and r.cin % 8 == 0 and not (r.cin >= 256 and r.cout >= 256) \
                        and ((M + 127) // 128) * ((r.cout + 127) // 128) >= 128
                    args = dict(i=[r.x.ld, r.ldwf, r.residual.ld if r.residual else 0, r.out.ld, r.x.B, r.x.H, r.x.W,
                                   r.cin, r.cout, r.k, r.s, int(bn is not None), int(r.upsample)], lane=ln)
                    # layers with few pixels (batch-1 inference): K split over the waves of a workgroup, no split-K slabs / fences
                    # (measured at batch 1, 640x640, end to end / device ms with the uint8 image: off 1.283 / 1.227, M <= 8192 0.955 / 0.842,
                    # M <= 32768 0.954 / 0.845, every layer 0.924 / 0.842 -- the 160x160 layers are a wash, larger M belongs to the
                    # throughput kernels)
                    lat_max = 32768
                    lat_e = fast and aligned and not wino_e and not pw_e and M <= lat_max and r.cin % 8 == 0 \
                        and bool(lib.yh_conv_lat_ok(r.x.B, r.x.H, r.x.W, r.cin, r.cout, r.k, r.s))
                    if lat_e:
                        r.wf = torch.zeros(kk * r.cin * r.ldwf, **f32)
                        latpacks.append((wfold.data_ptr(), r.wf.data_ptr(), r.cout, cin_real, kk, r.ldwf))
                        fwd.append(_op(L.OP_CONV_LAT_FWD_FUSED,
                                       p=[r.x.ptr(), r.wf, fbias, r.residual.ptr() if r.residual else None, r.out.ptr()], **args))
                    elif wino_e:
                        r.wf = torch.empty(16 * r.cin * r.ldwf, **f32)
                        winos.append((wfold.data_ptr(), r.wf.data_ptr(), r.cout, cin_real, r.ldwf, 0))
                        fwd.append(_op(L.OP_CONV_WINO_FWD_FUSED,
                                       p=[r.x.ptr(), r.wf, fbias, r.residual.ptr() if r.residual else None, r.out.ptr()], **args))
                    elif pw_e:
                        r.wf = torch.zeros(r.cin * r.ldwf, **f32)
                        pwpacks.append((wfold.data_ptr(), r.wf.data_ptr(), 0, r.cout, r.cin, r.ldwf, r.ldwb, 0, 0))
                        fwd.append(_op(L.OP_CONV_PW_FWD_FUSED,
                                       p=[r.x.ptr(), r.wf, fbias, r.residual.ptr() if r.residual else None, r.out.ptr()], **args))
                    else:
                        packs.append((wfold.data_ptr(), r.wf.data_ptr(), 0, r.cout, cin_real, kk, r.cin, r.ldwf, r.ldwb))
                        nws = int(lib.yh_conv_fwd_fused_ws(r.x.B, r.x.H, r.x.W, r.cin, r.cout, r.k, r.s))
                        sws = torch.zeros(nws, **f32) if nws > 0 else None   # per layer (two lanes may run split layers at once); zero tickets
                        if sws is not None:
                            keep.append(sws)
                        fwd.append(_op(L.OP_CONV_FWD_FUSED,
                                       p=[r.x.ptr(), r.wf, fbias, r.residual.ptr() if r.residual else None, r.out.ptr(), sws],
                                       l=[nws], **args))
                    continue
                if r.pair is not None and r.need_dx:
                    if r.pair_first:      # stacked backward packs: rows [0, c1) this conv, [c1, c1 + c2) its sibling
                        stacked = torch.empty((r.cout + r.pair.cout) * r.ldwb, **f32)
                        if r.pw_b:        # one k-quad interleaved matrix for both: each record writes its own K rows
                            r.wb = r.pair.wb = stacked
                        else:
                            r.wb, r.pair.wb = stacked[: r.cout * r.ldwb], stacked[r.cout * r.ldwb:]
                        keep.append(stacked)
                elif r.s2m_b:
                    r.ldwb = _rup4(2 * r.cin)
                    r.wb = torch.empty(6 * r.cout * r.ldwb, **f32)
                    s2m_packs.append(_op(L.OP_PACK_WEIGHTS_S2M, p=[r.weight, r.wb], i=[r.cout, r.cin, r.ldwb]))
                else:
                    # K rows of the backward pack: a 1x1 head reads dY rows zero-padded to a multiple of 4 channels against zero rows here
                    r.kcout_b = _rup4(r.cout) if (r.bn is None and r.k == 1 and not (r.wino_b or r.pw_b) and r.out.ldg >= _rup4(r.cout)) else r.cout
                    r.wb = (torch.zeros if r.kcout_b != r.cout else torch.empty)((16 if r.wino_b else kk) * r.kcout_b * r.ldwb, **f32) \
                        if r.need_dx else None
                second = r.pair is not None and not r.pair_first
                assert not r.fwd2 or r.pw_f                  # same eligibility rule (decided at pair detection)
                if r.pw_f:                                   # the pointwise pack writes only the conv's own columns
                    if r.fwd2 and not second:
                        r.ldwf = r.pair.ldwf = _rup4(r.cout + r.pair.cout)
                        r.wf = r.pair.wf = torch.zeros(r.cin * r.ldwf, **f32)
                    elif not r.fwd2:
                        r.wf = torch.zeros(r.cin * r.ldwf, **f32)
                if r.pw_f or r.pw_b:
                    koff = r.pair.cout if second else 0
                    pwpacks.append((r.weight.data_ptr(), r.wf.data_ptr() if r.pw_f else 0, r.wb.data_ptr() if r.pw_b else 0,
                                    r.cout, r.cin, r.ldwf, r.ldwb, koff, r.pair.cout if (second and r.fwd2) else 0))
                if r.wino_f:
                    r.wf = torch.empty(16 * r.cin * r.ldwf, **f32)
                    winos.append((r.weight.data_ptr(), r.wf.data_ptr(), r.cout, r.weight.shape[1], r.ldwf, 0))
                if r.wino_b:
                    winos.append((r.weight.data_ptr(), r.wb.data_ptr(), r.cout, r.weight.shape[1], r.ldwb, 1))
                if r.s2l_f:
                    r.wf = torch.zeros(kk * r.cin * r.ldwf, **f32)
                    latpacks.append((r.weight.data_ptr(), r.wf.data_ptr(), r.cout, r.weight.shape[1], kk, r.ldwf))
                gen_f = not (r.wino_f or r.pw_f or r.s2l_f)            # layouts the generic pack kernel still has to write
                gen_b = r.wb is not None and not (r.wino_b or r.pw_b or r.s2m_b)
                if gen_f or gen_b:
                    packs.append((r.weight.data_ptr(), r.wf.data_ptr() if gen_f else 0, r.wb.data_ptr() if gen_b else 0,
                                  r.cout, r.weight.shape[1], kk, r.cin, r.ldwf, r.ldwb))
                M = r.x.B * r.Ho * r.Wo

                def alloc_out(c: ConvRec, nblk: int):        # raw conv output, BN coefficients and partial-sum scratch
                    c.nblk = nblk
                    if c.bn is not None and c.coef is None:
                        c.y = torch.empty(c.x.B, c.Ho, c.Wo, c.cout, **f32)
                        c.dyp, c.lddy = c.y.data_ptr(), c.cout
                        if c.virtual:                         # the raw output lives in the output view itself (never normalised in memory)
                            c.yp, c.ldy = c.out.ptr(), c.out.ld
                        else:
                            c.yp, c.ldy = c.dyp, c.lddy
                        c.coef = torch.empty(4 * c.cout, **f32)
                        c.part = torch.empty(max(nblk, lib.yh_bn_bwd_blocks(M, c.cout)) * 2 * c.cout, **f32)

                if r.fwd2:
                    if r.pair_first:      # both siblings in one launch, at the first one's position in the list
                        nblk = lib.yh_conv_pw_blocks(M, r.cin, r.cout + r.pair.cout)
                        alloc_out(r, nblk)
                        alloc_out(r.pair, nblk)
                        q = r.pair
                        # on the main lane whatever lane the record was traced on: both siblings' BN passes depend on it
                        fwd.append(_op(L.OP_CONV_PW_FWD2, p=[r.x.ptr(), r.wf, r.bias, r.yp, r.part, q.bias, q.yp, q.part],
                                       i=[r.x.ld, r.ldwf, r.ldy, r.x.B, r.x.H, r.x.W, r.cin, r.cout, q.ldy, q.cout], lane=0))
                        if r.x_fused:
                            _set_prologue(fwd[-1], r.x)
                        if deferred_fork:
                            fwd.append(_op(L.OP_FORK))
                            deferred_fork = False
                        elif ln == 1 or q.lane == 1:
                            raise NotImplementedError("fused sibling convolution traced on the side lane without a preceding fork")
                else:
                    nblk = lib.yh_conv_narrow_blocks(r.x.B, r.x.H, r.x.W, r.cin, r.s) if r.narrow_f else \
                        lib.yh_conv_s2_blocks(r.x.B, r.x.H, r.x.W, r.cout) if r.s2l_f else \
                        lib.yh_conv_wino_blocks(r.x.B, r.x.H, r.x.W) if r.wino_f else \
                        lib.yh_conv_pw_blocks(M, r.cin, r.cout) if r.pw_f else \
                        lib.yh_conv_fwd_blocks(r.x.B, r.x.H, r.x.W, r.cout, r.k, r.s)
                    alloc_out(r, nblk)
                    ytarget, ldy = (r.yp, r.ldy) if r.bn is not None else (None, r.out.ld)
                    if r.narrow_f:
                        fwd.append(_op(L.OP_CONV_NARROW,
                                       p=[r.x.ptr(), r.wf, r.bias, ytarget if ytarget is not None else r.out.ptr(),
                                          r.part if r.bn is not None else None],
                                       i=[r.x.ld, r.ldwf, ldy, r.x.B, r.x.H, r.x.W, r.cin, r.cout, r.s, 0, 0], lane=ln))
                    else:
                        fwd.append(_op(L.OP_CONV_S2_FWD if r.s2l_f else L.OP_CONV_WINO_FWD if r.wino_f else (L.OP_CONV_PW_FWD if r.pw_f else L.OP_CONV_FWD),
                                       p=[r.x.ptr(), r.wf, r.bias, ytarget if ytarget is not None else r.out.ptr(),
                                          r.part if r.bn is not None else None],
                                       i=[r.x.ld, r.ldwf, ldy, r.x.B, r.x.H, r.x.W, r.cin, r.cout, r.k, r.s], lane=ln))
                    if r.x_fused:                             # the input's BatchNorm + SiLU is applied while this kernel stages it
                        _set_prologue(fwd[-1], r.x)
                nblk = r.nblk
                if r.bn is not None:
                    track = r.bn.track_running_stats and r.bn.running_mean is not None
                    mom = r.bn.momentum if r.bn.momentum is not None else 0.1
                    fwd.append(_op(L.OP_BN_FINALIZE,
                                   p=[r.part, r.bn.weight, r.bn.bias, r.bn.running_mean if track else None,
                                      r.bn.running_var if track else None, r.coef,
                                      r.bn.num_batches_tracked if track else None,
                                      r.out.icoef_ptr(0) if r.virtual else None, r.out.icoef_ptr(1) if r.virtual else None],
                                   i=[nblk, r.cout], f=[mom, r.bn.eps], l=[M], lane=ln))
                    if not r.virtual:
                        rf = self._res_fused(r)               # the residual is a raw conv output: normalised on the fly
                        fwd.append(_op(L.OP_BN_SILU_FWD,
                                       p=[r.yp, r.coef, r.residual.ptr() if r.residual else None, r.out.ptr(),
                                          r.residual.icoef_ptr() if rf else None],
                                       i=[r.ldy, r.residual.ld if r.residual else 0, r.out.ld, r.cout, r.Ho, r.Wo,
                                          int(r.upsample), r.residual.icoef_ld if rf else 0], l=[M], lane=ln))
            else:
                if r.fused:
                    continue
                if not self.training and lib.yh_sppf_pool3_ok(r.x.H, r.x.W):
                    # eval: SPPF's cascade pool5(pool5(pool5(x))) as one launch (no argmax, the plane stays in LDS)
                    nxt = self.recs[ri + 1: ri + 3]
                    if len(nxt) == 2 and all(isinstance(q, PoolRec) and q.lane == ln for q in nxt) \
                            and _same_view(nxt[0].x, r.out) and _same_view(nxt[1].x, nxt[0].out) \
                            and r.out.ld == nxt[0].out.ld == nxt[1].out.ld \
                            and all(v.ptr() % 16 == 0 for v in (r.out, nxt[0].out, nxt[1].out)):
                        fwd.append(_op(L.OP_SPPF_POOL3, p=[r.x.ptr(), r.out.ptr(), nxt[0].out.ptr(), nxt[1].out.ptr()],
                                       i=[r.x.ld, r.out.ld, r.x.B, r.x.H, r.x.W, r.x.C], lane=ln))
                        nxt[0].fused = nxt[1].fused = True
                        continue
                r.arg = torch.empty(r.x.B, r.x.H, r.x.W, r.x.C, device=dev, dtype=torch.uint8)
                fwd.append(_op(L.OP_MAXPOOL5_FWD, p=[r.x.ptr(), r.out.ptr(), r.arg],
                               i=[r.x.ld, r.out.ld, r.x.B, r.x.H, r.x.W, r.x.C], lane=ln))
        import struct
        head: List[L.YhOp] = list(reversed(s2m_packs))     # weight transforms: head of the forward list (training) / prep list (eval)
        if pwpacks:
            blob = b"".join(struct.pack("<QQQiiiiii", *d) for d in pwpacks)
            self.pw_table = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
            head.insert(0, _op(L.OP_PW_PACK_MULTI, p=[self.pw_table], i=[len(pwpacks)]))
        if latpacks:
            blob = b"".join(struct.pack("<QQiiii", *d) for d in latpacks)
            self.lat_table = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
            head.insert(0, _op(L.OP_LAT_PACK_MULTI, p=[self.lat_table], i=[len(latpacks)]))
        if winos:
            blob = b"".join(struct.pack("<QQiiii", *d) for d in winos)
            self.wino_table = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
            head.insert(0, _op(L.OP_WINO_WEIGHTS_MULTI, p=[self.wino_table], i=[len(winos)]))
        if packs:   # every conv's OIHW -> packed copies in ONE launch
            blob = b"".join(struct.pack("<QQQiiiiiiii", *d, 0, 0) for d in packs)
            self.pack_table = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
            head.insert(0, _op(L.OP_PACK_WEIGHTS_MULTI, p=[self.pack_table], i=[len(packs)]))
        if folds:
            # eval: BatchNorm folding + the transforms of the folded weights read only parameters and running statistics:
            # they run once per weight state (refresh_folded_weights), not once per forward -- at batch 1 the fold was the
            # longest launch of the pass
            blob = b"".join(struct.pack("<QQQQQQQQiifi", *d) for d in folds)
            self.fold_table = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
            head.insert(0, _op(L.OP_FOLD_OIHW_MULTI, p=[self.fold_table], i=[len(folds)]))
            self.prep_ops = _pack(head)
        else:
            fwd = head + fwd
        self.fwd_ops = _pack(fwd)
        self.bwd_ops = _pack(self._lower_backward(grad_of, keep)) if self.training else None
        self._keep = keep
        self._grad_of = grad_of
        self.param_ptrs = self._signature()

    # ---- bf16 lowering (BASELINE configs 3-4) -------------------------------------------------------------------------
    def _compile_bf16(self, grad_of):
        """One kernel family for every convolution: yh_bf16_conv_fwd / _bwd_data / _bwd_weight (bf16 operands, fp32
        accumulation), bf16 BN+SiLU / pool passes, fp32 statistics and parameter gradients."""
        import struct
        lib = L.lib()
        dev = self.device
        f32 = dict(device=dev, dtype=torch.float32)
        b16 = dict(device=dev, dtype=torch.bfloat16)
        fwd: List[L.YhOp] = []
        keep: List[torch.Tensor] = []
        packs: List[tuple] = []
        # sibling pointwise convs (C3 conv1 / conv2): one backward-data GEMM over K = Cout1 + Cout2, dx written once
        if True:
            groups: Dict[tuple, List[ConvRec]] = {}
            for r in self.recs:
                if isinstance(r, ConvRec) and r.k == 1 and r.s == 1 and r.bn is not None and r.cout % 8 == 0:
                    groups.setdefault((id(r.x.buf), r.x.off, r.x.C), []).append(r)
            for grp in groups.values():
                if len(grp) == 2 and grp[0].cout == grp[1].cout and (self.need_input_grad or grp[0].x.buf is not self.input):
                    grp[0].pair, grp[1].pair = grp[1], grp[0]
                    grp[0].pair_first = True
        for r in self.recs:
            if isinstance(r, SyncRec):
                fwd.append(_op(L.OP_FORK if r.kind == "fork" else L.OP_JOIN))
                continue
            ln = r.lane
            if isinstance(r, PoolRec):
                r.arg = torch.empty(r.x.B, r.x.H, r.x.W, r.x.C, device=dev, dtype=torch.uint8)
                fwd.append(_op(L.OP_BF16_MAXPOOL5_FWD, p=[r.x.ptr(), r.out.ptr(), r.arg],
                               i=[r.x.ld, r.out.ld, r.x.B, r.x.H, r.x.W, r.x.C], lane=ln))
                continue
            kk = r.k * r.k
            cin_real = r.weight.shape[1]
            r.need_dx = self.need_input_grad or r.x.buf is not self.input
            r.ldwf, r.ldwb = _rup8(r.cout), _rup8(r.cin)
            r.wf = torch.empty(kk * r.cin * r.ldwf, **b16)
            kpad = _rup8(r.cout)
            koff = 0
            if r.need_dx:
                if r.pair is not None:
                    kpad = r.cout + r.pair.cout
                    if r.pair_first:
                        r.wb = r.pair.wb = torch.zeros(kpad * r.ldwb, **b16)
                    else:
                        koff = r.pair.cout
                else:
                    r.wb = torch.zeros(kk * kpad * r.ldwb, **b16)
            packs.append((r.weight.data_ptr(), r.wf.data_ptr(), r.wb.data_ptr() if r.need_dx else 0, r.cout, cin_real, kk,
                          r.cin, r.ldwf, r.ldwb, koff, kpad))
            M = r.x.B * r.Ho * r.Wo
            r.nblk = lib.yh_bf16_conv_fwd_blocks(r.x.B, r.x.H, r.x.W, r.cin, r.cout, r.k, r.s, 0 if r.bn is not None else 1, r.x.ld,
                                                 r.cout if r.bn is not None else r.out.ld)
            # narrow high-resolution 3x3 layers (first layer, stem[3], the 16-channel bottleneck): the direct kernels of the fp32
            # path with bf16 storage (conv_narrow.hip); cin_k = channels the kernel reads (4 of the first layer's 8 padded ones)
            use_nar = r.k == 3 and r.bn is not None and r.x.ld % 4 == 0 and r.x.off % 4 == 0
            r.cin_k = 4 if (cin_real <= 4 and r.cin == 8) else r.cin
            # measured at batch 64 (tools/layer_bench.py --dtype bf16, ms narrow / generic bf16): first layer forward 0.250 /
            # 0.358 and weight gradient 0.220 / 0.541; stem[3] forward 0.250 / 0.114, backward-data 0.295 / 0.338, weight gradient
            # 0.274 / 0.155; 16 -> 16 forward 0.098 / 0.098, backward-data 0.114 / 0.099, weight gradient 0.113 / 0.129 per layer
            # (8-byte pieces: twice the load instructions per byte of the fp32 form -- these kernels are bound by instruction
            # issue and latency, not by bytes).  Each pass takes the faster kernel; YH_BF16_NARROW=all forces the narrow ones.
            every = False
            # round 3: stem[3] forward (16 -> 32, stride 2) has a bf16-MFMA form of the narrow kernel (0.112 -> 0.075 ms)
            stem3 = r.s == 2 and r.cin == 16 and r.cout == 32 and r.cin == cin_real and r.x.ld % 8 == 0
            r.narrow_f = bool(use_nar and (r.cin_k == 4 or stem3 or (every and r.cin == cin_real)) and lib.yh_conv_narrow_ok(r.cin_k, r.cout, 3, r.s))
            r.narrow_w = bool(use_nar and (r.cin_k == 4 or r.s == 1 or every) and
                              lib.yh_conv_narrow_bwd_weight_ok(r.cin_k, min(cin_real, r.cin_k), r.cout, 3, r.s))
            r.narrow_b = bool(use_nar and r.need_dx and r.pair is None and r.cin == cin_real and
                              ((r.s == 1 and every and lib.yh_conv_narrow_ok(r.cout, r.cin, 3, 1)) or
                               (r.s == 2 and lib.yh_conv_narrow_dgrad_s2_ok(r.cin, r.cout))))
            if r.narrow_f:
                r.nblk = lib.yh_conv_narrow_blocks(r.x.B, r.x.H, r.x.W, r.cin_k, r.s)
            if r.bn is not None:
                r.y = torch.empty(r.x.B, r.Ho, r.Wo, r.cout, **b16)
                r.coef = torch.empty(4 * r.cout, **f32)
                r.part = torch.empty(max(r.nblk, lib.yh_bn_bwd_blocks(M, r.cout)) * 2 * r.cout, **f32)
                if r.narrow_f:
                    fwd.append(_op(L.OP_BF16_CONV_NARROW, p=[r.x.ptr(), r.wf, r.bias, r.y, r.part],
                                   i=[r.x.ld, r.ldwf, r.cout, r.x.B, r.x.H, r.x.W, r.cin_k, r.cout, r.s, 0, 0, r.cin], lane=ln))
                else:
                    fwd.append(_op(L.OP_BF16_CONV_FWD, p=[r.x.ptr(), r.wf, r.bias, r.y, r.part],
                                   i=[r.x.ld, r.ldwf, r.cout, r.x.B, r.x.H, r.x.W, r.cin, r.cout, r.k, r.s, 0], lane=ln))
                track = r.bn.track_running_stats and r.bn.running_mean is not None
                mom = r.bn.momentum if r.bn.momentum is not None else 0.1
                fwd.append(_op(L.OP_BN_FINALIZE,
                               p=[r.part, r.bn.weight, r.bn.bias, r.bn.running_mean if track else None,
                                  r.bn.running_var if track else None, r.coef, r.bn.num_batches_tracked if track else None,
                                  None, None],
                               i=[r.nblk, r.cout], f=[mom, r.bn.eps], l=[M], lane=ln))
                fwd.append(_op(L.OP_BF16_BN_SILU_FWD,
                               p=[r.y, r.coef, r.residual.ptr() if r.residual else None, r.out.ptr()],
                               i=[r.cout, r.residual.ld if r.residual else 0, r.out.ld, r.cout, r.Ho, r.Wo, int(r.upsample)],
                               l=[M], lane=ln))
            else:       # head output: fp32, no BatchNorm
                fwd.append(_op(L.OP_BF16_CONV_FWD, p=[r.x.ptr(), r.wf, r.bias, r.out.ptr(), None],
                               i=[r.x.ld, r.ldwf, r.out.ld, r.x.B, r.x.H, r.x.W, r.cin, r.cout, r.k, r.s, 1], lane=ln))
        blob = b"".join(struct.pack("<QQQiiiiiiii", *d) for d in packs)
        self.pack_table = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
        fwd.insert(0, _op(L.OP_BF16_PACK_MULTI, p=[self.pack_table], i=[len(packs)]))
        self.fwd_ops = _pack(fwd)
        self.bwd_ops = _pack(self._lower_backward_bf16(grad_of))
        self._keep = keep
        self._grad_of = grad_of
        self.param_ptrs = self._signature()

    def _lower_backward_bf16(self, grad_of) -> List[L.YhOp]:
        lib = L.lib()
        if grad_of is None:
            raise ValueError("training plan needs gradient destinations")
        for b in self.buffers:
            b.grad_cover = []
        for v, _ in self.outputs:
            v.buf.grad_cover.append((v.off, v.off + v.C))
        ws_floats = 1
        for r in self.recs:
            if isinstance(r, ConvRec):
                ws_floats = max(ws_floats, lib.yh_conv_narrow_bwd_weight_ws(r.x.B, r.x.H, r.x.W, r.cin_k, r.cout, r.s) if r.narrow_w else
                                lib.yh_bf16_conv_bwd_weight_ws(r.x.B, r.x.H, r.x.W, r.cin, r.cout, r.k, r.s))
                if r.bias is not None:
                    ws_floats = max(ws_floats, lib.yh_colsum_ws(r.x.B * r.Ho * r.Wo, r.cout))
        self.ws = torch.empty(int(ws_floats), device=self.device, dtype=torch.float32)
        ops: List[L.YhOp] = []
        pair_pending = set()
        self.grad_ready = {}
        for r in reversed(self.recs):
            if isinstance(r, SyncRec):
                continue
            if isinstance(r, PoolRec):
                dst, acc = self._grad_target(r.x)
                if not acc:
                    raise NotImplementedError("max-pool backward expects an already written input gradient")
                ops.append(_op(L.OP_BF16_MAXPOOL5_BWD, p=[r.out.gptr(), r.arg, dst], i=[r.out.ldg, r.x.ldg, r.x.B, r.x.H, r.x.W, r.x.C]))
                continue
            M = r.x.B * r.Ho * r.Wo
            if r.bn is not None:
                nb = lib.yh_bn_bwd_blocks(M, r.cout)
                ops.append(_op(L.OP_BF16_BN_SILU_BWD_REDUCE, p=[r.out.gptr(), r.y, r.coef, r.part],
                               i=[r.out.ldg, r.cout, r.cout, r.Ho, r.Wo, int(r.upsample)], l=[M]))
                if r.residual is not None:
                    dres, racc = self._grad_target(r.residual)
                    ldres = r.residual.ldg
                else:
                    dres, racc, ldres = None, 0, 0
                ops.append(_op(L.OP_BF16_BN_SILU_BWD_APPLY,
                               p=[r.out.gptr(), r.y, r.coef, r.part, r.bn.weight, grad_of[id(r.bn.weight)], grad_of[id(r.bn.bias)],
                                  r.y, dres],
                               i=[r.out.ldg, r.cout, nb, r.cout, ldres, racc, r.cout, r.Ho, r.Wo, int(r.upsample)], l=[M]))
                dy, lddy, kcout = r.y.data_ptr(), r.cout, r.cout
                self.grad_ready[id(r.bn.weight)] = self.grad_ready[id(r.bn.bias)] = len(ops)
            else:       # head: the loss wrote a bf16 gradient zero-padded to a multiple of 8 channels
                dy, lddy, kcout = r.out.gptr(), r.out.ldg, _rup8(r.cout)
                if r.out.off != 0 or r.out.ldg != kcout:
                    raise NotImplementedError("head gradient must own its (padded) buffer")
            fuse_bias = r.bias is not None and r.narrow_w and r.bn is not None
            if r.bias is not None and not fuse_bias:
                # head biases (18 / 255 channels): sum the zero-padded multiple of 4 so the vector kernel applies; the extra
                # column sums are exact zeros and land in the 4-float padding every tensor has in the flat gradient buffer
                cs = _rup4(r.cout) if (r.bn is None and _rup4(r.cout) <= lddy) else r.cout
                ops.append(_op(L.OP_BF16_COLSUM, p=[dy, grad_of[id(r.bias)], self.ws], i=[lddy, cs], l=[M]))
                self.grad_ready[id(r.bias)] = len(ops)
            if r.narrow_w:
                ops.append(_op(L.OP_BF16_CONV_NARROW_BWD_WEIGHT,
                               p=[r.x.ptr(), dy, grad_of[id(r.weight)], self.ws, grad_of[id(r.bias)] if fuse_bias else None],
                               i=[r.x.ld, lddy, r.x.B, r.x.H, r.x.W, r.cin_k, min(r.weight.shape[1], r.cin_k), r.cout, r.k, r.s],
                               l=[self.ws.numel()]))
            else:
                ops.append(_op(L.OP_BF16_CONV_BWD_WEIGHT, p=[r.x.ptr(), dy, grad_of[id(r.weight)], self.ws],
                               i=[r.x.ld, lddy, r.x.B, r.x.H, r.x.W, r.cin, r.weight.shape[1], r.cout, r.k, r.s], l=[self.ws.numel()]))
            self.grad_ready[id(r.weight)] = len(ops)
            if fuse_bias:
                self.grad_ready[id(r.bias)] = len(ops)
            if not r.need_dx:
                continue
            if r.pair is not None:
                if id(r.pair) in pair_pending:      # second of the pair in backward order: both dY are final now
                    first, second = (r, r.pair) if r.pair_first else (r.pair, r)
                    dst, acc = self._grad_target(r.x)
                    ops.append(_op(L.OP_BF16_CONV_BWD_DATA, p=[first.y, first.wb, dst, second.y],
                                   i=[first.cout, first.ldwb, r.x.ldg, r.x.B, r.x.H, r.x.W, r.cin, first.cout + second.cout, 1, 1,
                                      acc, first.cout]))
                else:
                    pair_pending.add(id(r))
                continue
            dst, acc = self._grad_target(r.x)
            if r.narrow_b and r.s == 1:      # x = dY (Cout channels), output = dX (Cin channels), backward pack, flipped taps
                ops.append(_op(L.OP_BF16_CONV_NARROW, p=[dy, r.wb, None, dst, None],
                               i=[lddy, r.ldwb, r.x.ldg, r.x.B, r.x.H, r.x.W, r.cout, r.cin, 1, 1, acc, _rup8(r.cout)]))
            elif r.narrow_b:
                ops.append(_op(L.OP_BF16_CONV_NARROW_DGRAD_S2, p=[dy, r.wb, dst],
                               i=[lddy, r.ldwb, r.x.ldg, r.x.B, r.x.H, r.x.W, r.cin, r.cout, r.k, r.s, acc, _rup8(r.cout)]))
            else:
                ops.append(_op(L.OP_BF16_CONV_BWD_DATA, p=[dy, r.wb, dst, None],
                               i=[lddy, r.ldwb, r.x.ldg, r.x.B, r.x.H, r.x.W, r.cin, kcout, r.k, r.s, acc, 0]))
        return ops

    def _grad_target(self, v: View) -> Tuple[int, int]:
        """Pointer into the gradient tensor of `v` and whether the op must accumulate."""
        cov = v.buf.grad_cover
        lo, hi = v.off, v.off + v.C
        inside = any(a <= lo and hi <= b for a, b in cov)
        overlap = any(a < hi and lo < b for a, b in cov)
        if inside:
            return v.gptr(), 1
        if overlap:
            raise NotImplementedError(f"partial gradient overlap on buffer {v.buf.name}")
        cov.append((lo, hi))
        return v.gptr(), 0

    def _lower_backward(self, grad_of, keep) -> List[L.YhOp]:
        lib = L.lib()
        if grad_of is None:
            raise ValueError("training plan needs gradient destinations")
        for b in self.buffers:
            b.grad_cover = []
        # gradients of the declared outputs arrive from outside (loss kernel / autograd) fully written
        for v, _ in self.outputs:
            v.buf.grad_cover.append((v.off, v.off + v.C))
        ws_floats = 1
        for r in self.recs:
            if isinstance(r, ConvRec):
                ws_floats = max(ws_floats, lib.yh_conv_narrow_bwd_weight_ws(r.x.B, r.x.H, r.x.W, r.cin, r.cout, r.s) if r.narrow_w else
                                lib.yh_conv_wino_bwd_weight_ws(r.x.B, r.x.H, r.x.W, r.cin, r.cout) if r.wino_w else
                                lib.yh_conv_pw_bwd_weight_ws(r.x.B * r.x.H * r.x.W, r.cin, r.cout) if r.pw_w else
                                lib.yh_conv_bwd_weight_ws(r.x.B, r.x.H, r.x.W, r.cin, r.cout, r.k, r.s))
                if r.bias is not None:
                    ws_floats = max(ws_floats, lib.yh_colsum_ws(r.x.B * r.Ho * r.Wo, r.cout))
        self.ws = torch.empty(int(ws_floats), device=self.device, dtype=torch.float32)
        ops: List[L.YhOp] = []
        pair_pending = set()
        self.grad_ready: Dict[int, int] = {}     # id(param) -> number of backward ops after which its grad is final
        writers: List[tuple] = []                # (op index, view written, consumer rec | None): every op that writes an activation gradient
        bn_ops: Dict[int, tuple] = {}            # id(rec) -> (rec, index of its REDUCE op, index of its APPLY op)
        for r in reversed(self.recs):
            if isinstance(r, SyncRec):
                continue
            if isinstance(r, PoolRec):
                dst, acc = self._grad_target(r.x)
                if not acc:
                    raise NotImplementedError("max-pool backward expects an already written input gradient")
                ops.append(_op(L.OP_MAXPOOL5_BWD, p=[r.out.gptr(), r.arg, dst], i=[r.out.ld, r.x.ld, r.x.B, r.x.H, r.x.W, r.x.C]))
                writers.append((len(ops) - 1, r.x, None))
                continue
            M = r.x.B * r.Ho * r.Wo
            if r.bn is not None:
                nb = lib.yh_bn_bwd_blocks(M, r.cout)
                ops.append(_op(L.OP_BN_SILU_BWD_REDUCE, p=[r.out.gptr(), r.yp, r.coef, r.part],
                               i=[r.out.ld, r.ldy, r.cout, r.Ho, r.Wo, int(r.upsample)], l=[M]))
                if r.residual is not None:
                    dres, racc = self._grad_target(r.residual)
                    ldres = r.residual.ld
                else:
                    dres, racc, ldres = None, 0, 0
                ops.append(_op(L.OP_BN_SILU_BWD_APPLY,
                               p=[r.out.gptr(), r.yp, r.coef, r.part, r.bn.weight, grad_of[id(r.bn.weight)],
                                  grad_of[id(r.bn.bias)], r.dyp, dres],
                               i=[r.out.ld, r.ldy, nb, r.lddy, ldres, racc, r.cout, r.Ho, r.Wo, int(r.upsample)], l=[M]))
                bn_ops[id(r)] = (r, len(ops) - 2, len(ops) - 1)
                if r.residual is not None:
                    writers.append((len(ops) - 1, r.residual, None))
                dy, lddy = r.dyp, r.lddy
                self.grad_ready[id(r.bn.weight)] = self.grad_ready[id(r.bn.bias)] = len(ops)
            else:
                dy, lddy = r.out.gptr(), r.out.ldg                 # head: the loss / autograd wrote rows zero-padded to ldg channels
            kcout = (r.kcout_b or r.cout) if r.bn is None else r.cout
            fuse_bias = r.bias is not None and r.narrow_w          # the narrow weight-gradient kernel also sums dY's columns
            if r.bias is not None and not fuse_bias:
                # head biases: sum the zero-padded multiple of 4 so the vector kernel applies; the extra column sums are exact zeros
                # and land in the 4-float padding every tensor has in the flat gradient buffer
                cs = _rup4(r.cout) if (r.bn is None and _rup4(r.cout) <= lddy) else r.cout
                ops.append(_op(L.OP_COLSUM, p=[dy, grad_of[id(r.bias)], self.ws], i=[lddy, cs], l=[M]))
                self.grad_ready[id(r.bias)] = len(ops)
            ops.append(_op(L.OP_CONV_NARROW_BWD_WEIGHT if r.narrow_w else L.OP_CONV_WINO_BWD_WEIGHT if r.wino_w else
                           (L.OP_CONV_PW_BWD_WEIGHT if r.pw_w else L.OP_CONV_BWD_WEIGHT),
                           p=[r.x.ptr(), dy, grad_of[id(r.weight)], self.ws, grad_of[id(r.bias)] if fuse_bias else None],
                           i=[r.x.ld, lddy, r.x.B, r.x.H, r.x.W, r.cin, r.weight.shape[1], r.cout, r.k, r.s],
                           l=[self.ws.numel()]))
            if r.x_fused:                                          # x is a raw conv output: the kernel normalises it while staging
                _set_prologue(ops[-1], r.x)
            self.grad_ready[id(r.weight)] = len(ops)
            if fuse_bias:
                self.grad_ready[id(r.bias)] = len(ops)
            if r.need_dx and r.pair is not None:
                if id(r.pair) in pair_pending:      # second of the pair (in backward order): both dY are final now
                    first, second = (r, r.pair) if r.pair_first else (r.pair, r)
                    dst, acc = self._grad_target(r.x)
                    if r.pw_b:
                        ops.append(_op(L.OP_CONV_PW_BWD_DATA, p=[first.y, second.y, first.wb, dst],
                                       i=[first.cout, second.cout, first.cout, first.ldwb, r.x.ld, r.x.B, r.x.H, r.x.W, r.cin, acc]))
                    else:
                        ops.append(_op(L.OP_CONV_BWD_DATA_PAIR, p=[first.y, second.y, first.wb, dst],
                                       i=[first.cout, second.cout, first.cout, first.ldwb, r.x.ld, r.x.B, r.x.H, r.x.W, r.cin, acc]))
                    writers.append((len(ops) - 1, r.x, r))
                else:
                    pair_pending.add(id(r))
            elif r.need_dx:
                dst, acc = self._grad_target(r.x)
                writers.append((len(ops), r.x, r))
                if r.narrow_b and r.s == 2:
                    ops.append(_op(L.OP_CONV_NARROW_DGRAD_S2, p=[dy, r.wb, dst],
                                   i=[lddy, r.ldwb, r.x.ld, r.x.B, r.x.H, r.x.W, r.cin, r.cout, r.k, r.s, acc]))
                elif r.narrow_b:     # x = dY (Cout channels), output = dX (Cin channels), backward pack, flipped taps
                    ops.append(_op(L.OP_CONV_NARROW, p=[dy, r.wb, None, dst, None],
                                   i=[lddy, r.ldwb, r.x.ld, r.x.B, r.x.H, r.x.W, r.cout, r.cin, 1, 1, acc]))
                elif r.s2m_b:
                    ops.append(_op(L.OP_CONV_BWD_DATA_S2M, p=[dy, r.wb, dst],
                                   i=[lddy, r.ldwb, r.x.ld, r.x.B, r.x.H, r.x.W, r.cin, r.cout, r.k, r.s, acc]))
                elif r.pw_b:
                    ops.append(_op(L.OP_CONV_PW_BWD_DATA, p=[dy, None, r.wb, dst],
                                   i=[r.cout, 0, lddy, r.ldwb, r.x.ld, r.x.B, r.x.H, r.x.W, r.cin, acc]))
                elif r.wino_b:
                    ops.append(_op(L.OP_CONV_WINO_BWD_DATA, p=[dy, r.wb, dst],
                                   i=[lddy, r.ldwb, r.x.ld, r.x.B, r.x.H, r.x.W, r.cin, r.cout, acc]))
                else:
                    ops.append(_op(L.OP_CONV_BWD_DATA, p=[dy, r.wb, dst],
                                   i=[lddy, r.ldwb, r.x.ld, r.x.B, r.x.H, r.x.W, r.cin, kcout, r.k, r.s, acc]))
        # (folding the BatchNorm-backward sums into the epilogue of the backward-data GEMM that finishes a layer's gradient was
        # built, tested and measured in round 2: backward-data 5.74 -> 7.26 ms against 0.85 ms saved in the reduce pass; retired)
        return ops

    # ---- execution ----------------------------------------------------------------------------
    def _signature(self) -> List[int]:
        """Every device address the op lists bake in that the plan does not own: parameters, BatchNorm buffers
        (running statistics, num_batches_tracked) and the gradient destinations."""
        if self._sig_holders is None:       # (module, attribute) pairs: `bn.running_mean = t` / `p.data = t` must be seen
            hold, seen = [], set()
            for r in self.recs:
                if not isinstance(r, ConvRec):
                    continue
                if id(r.conv) not in seen:
                    seen.add(id(r.conv))
                    hold += [(r.conv, "weight"), (r.conv, "bias")]
                if r.bn is not None and id(r.bn) not in seen:
                    seen.add(id(r.bn))
                    hold += [(r.bn, "weight"), (r.bn, "bias"), (r.bn, "running_mean"), (r.bn, "running_var"),
                             (r.bn, "num_batches_tracked")]
            self._sig_holders = hold
        sig: List[int] = []
        for obj, attr in self._sig_holders:
            t = getattr(obj, attr)
            if t is None:
                sig.append(0)
                continue
            sig.append(t.data_ptr())
            if self._grad_of:
                g = self._grad_of.get(id(t))
                sig.append(g.data_ptr() if g is not None else 0)
        return sig

    def params_moved(self) -> bool:
        """True when a parameter, BatchNorm buffer or gradient destination no longer lives at the address the op
        lists hold (HipTrainer moved the parameters into its flat buffer, load_state_dict(assign=True), a
        reassigned running_mean, ...): the plan must be re-traced, a captured hipGraph re-captured."""
        return self._signature() != self.param_ptrs

    def _ctx(self):
        """This thread's execution context for the plan's device (the C side refuses a context on a foreign device)."""
        return L.context_for(self.device.index if self.device.index is not None else torch.cuda.current_device())

    def _on_device(self):
        dev = self.device.index
        return torch.cuda.device(dev) if dev is not None and torch.cuda.current_device() != dev else _NullCtx()

    def weights_state(self):
        """What the folded inference weights were computed from: the process-wide counter of in-place updates made by this
        package's own kernels (which write through raw pointers and so bypass torch's version counters: optimizer steps,
        BatchNorm running statistics of training forwards) plus torch's version counter of every tensor the fold reads."""
        if self._fold_tensors is None:
            self._signature()
            self._fold_tensors = [(o, a) for o, a in self._sig_holders if a != "num_batches_tracked"]
        v = 0
        for o, a in self._fold_tensors:
            t = getattr(o, a)
            if t is not None:
                v += t._version
        return (WEIGHTS_EPOCH[0], v)

    def fold_inputs(self) -> List[torch.Tensor]:
        """The tensors the fold reads, resolved once (InferenceSession sums their version counters per image without
        walking the module tree; a re-assigned attribute shows up as a PARAM_GENERATION / signature change instead)."""
        self.weights_state()
        return [t for t in (getattr(o, a) for o, a in self._fold_tensors) if t is not None]

    def refresh_folded_weights(self, stream: int) -> bool:
        """Eval plans: (re)fold BatchNorm into the packed weights when the weights changed since the last fold."""
        if self.prep_ops is None:
            return False
        state = self.weights_state()
        if state == self._folded_state:
            return False
        with self._on_device():
            L.run_ops(self.prep_ops[0], self.prep_ops[1], stream, None)
        self._folded_state = state
        return True

    def run_forward(self, stream: int):
        if self.training:
            WEIGHTS_EPOCH[0] += 1          # running statistics / num_batches_tracked are about to change under torch's radar
        else:
            self.refresh_folded_weights(stream)
        with self._on_device():
            L.run_ops(self.fwd_ops[0], self.fwd_ops[1], stream, self._ctx())
        self.generation += 1

    def run_backward(self, stream: int, begin: int = 0, end: Optional[int] = None):
        """Run backward ops [begin, end) (the trainer splits the list at gradient-bucket boundaries)."""
        arr, n = self.bwd_ops
        end = n if end is None else end
        if end > begin:
            with self._on_device():
                L.run_ops(ctypes.cast(ctypes.byref(arr, begin * ctypes.sizeof(L.YhOp)), ctypes.POINTER(L.YhOp)), end - begin,
                          stream, self._ctx())


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def _addr(x) -> Optional[int]:
    if x is None:
        return None
    if isinstance(x, int):
        return x
    return x.data_ptr()


def _op(kind: int, p=(), i=(), f=(), l=(), lane: int = 0) -> L.YhOp:
    o = L.YhOp()
    o.kind = kind
    o.lane = lane
    for n, v in enumerate(p):
        o.p[n] = _addr(v)
    for n, v in enumerate(i):
        o.i[n] = int(v)
    for n, v in enumerate(f):
        o.f[n] = float(v)
    for n, v in enumerate(l):
        o.l[n] = int(v)
    return o


def _set_prologue(o: L.YhOp, x: "View"):
    """Convolution records carry the input-prologue table of their x operand in p[10] and its row stride in i[17]."""
    o.p[10] = x.icoef_ptr()
    o.i[17] = x.icoef_ld


def _pack(ops: List[L.YhOp]):
    arr = (L.YhOp * max(len(ops), 1))()
    for n, o in enumerate(ops):
        arr[n] = o
    return arr, len(ops)
