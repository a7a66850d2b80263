"""Pose-ResNet backbone -- drop-in for the reference's ``lib/models/resnet.py`` (``ResNet`` :98-201,
``get(cfg)`` :211-215): same constructor input (``cfg.RESNET.*``, ``cfg.DATASET.NUM_JOINTS``), same
``state_dict`` keys and shapes, ``forward(x [N,3,H,W]) -> heatmaps [N,J,H/4,W/4]``.

The module holds parameters only.  ``forward`` runs the whole network through the bf16 HIP
interpreter ``fvp_bb_run`` (NHWC bf16 activations, implicit-GEMM convs on
``v_mfma_f32_32x32x16_bf16`` with fp32 accumulation, eval BatchNorm folded to a scale / shift in the
epilogue together with the residual add and ReLU).  ``forward_channels_last`` returns the fp32
heatmaps directly in the ``[N, H*W, JP]`` staging layout the projection kernels read.
"""
import ctypes as C
import os

import torch

from .. import _capi as capi
from ..netspec import ParamTree

SPEC = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottleneck", [3, 4, 6, 3]),
        101: ("bottleneck", [3, 4, 23, 3]), 152: ("bottleneck", [3, 8, 36, 3])}
BN_EPS = 1e-5


def _up(x, m):
    return (x + m - 1) // m * m


class PoseResNet(ParamTree):
    def __init__(self, cfg, _lib=None):
        super().__init__()
        r = cfg.RESNET
        self.__dict__["block"], self.__dict__["layers"] = SPEC[int(r.NUM_LAYERS)]
        assert list(r.NUM_DECONV_KERNELS) == [4] * int(r.NUM_DECONV_LAYERS), "only ConvTranspose(k4, s2, p1) deconvs"
        assert int(r.FINAL_CONV_KERNEL) == 1
        self.__dict__["deconv_filters"] = [int(f) for f in r.NUM_DECONV_FILTERS]
        self.__dict__["deconv_bias"] = bool(r.DECONV_WITH_BIAS)
        self.__dict__["num_joints"] = int(cfg.DATASET.NUM_JOINTS)
        self.__dict__["device_name"] = str(cfg.DEVICE)
        dev = torch.device(cfg.DEVICE)
        if _lib is None and dev.type != "cuda":
            raise capi.FvpError("the backbone runs on the GPU only (no CPU fallback)")
        self.__dict__["lib"] = _lib if _lib is not None else capi.load()
        self.__dict__["_convs"] = []          # (conv key, bn key or None, op template)
        self._declare()
        self.__dict__["_dirty"] = True
        self.__dict__["_plans"] = {}
        # per-(image size, batch) choice of the conv tile configuration by measurement (fvp_bb_tune); set
        # ``model.autotune = False`` to keep the built-in heuristic
        self.__dict__["autotune"] = True
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.mark_dirty())

    # ---- parameters (registration order = the reference's module order) ------------------------------------------
    def _conv(self, key, cin, cout, k, transposed=False, bias=False):
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        self.add(key + ".weight", torch.zeros(shape))
        if bias:
            self.add(key + ".bias", torch.zeros(cout))

    def _bn(self, key, c):
        self.add(key + ".weight", torch.ones(c))
        self.add(key + ".bias", torch.zeros(c))
        self.add(key + ".running_mean", torch.zeros(c), buffer=True)
        self.add(key + ".running_var", torch.ones(c), buffer=True)
        self.add(key + ".num_batches_tracked", torch.zeros((), dtype=torch.long), buffer=True)

    def _declare(self):
        ops = self._convs
        self._conv("conv1", 3, 64, 7)
        self._bn("bn1", 64)
        ops.append(dict(kind=capi.BB_CONV, key="conv1", bn="bn1", cin=3, cout=64, k=7, stride=2, pad=3, relu=True,
                        src="x", dst="c1", stem=True))
        ops.append(dict(kind=capi.BB_MAXPOOL, cin=64, cout=64, src="c1", dst="p1"))
        cur, inplanes = "p1", 64
        exp = 4 if self.block == "bottleneck" else 1
        for li, (planes, nblocks) in enumerate(zip((64, 128, 256, 512), self.layers), start=1):
            for b in range(nblocks):
                pre = f"layer{li}.{b}"
                stride = 2 if (b == 0 and li > 1) else 1
                down = b == 0 and (stride != 1 or inplanes != planes * exp)
                if self.block == "bottleneck":
                    chain = [("conv1", "bn1", inplanes, planes, 1, 1, 0), ("conv2", "bn2", planes, planes, 3, stride, 1),
                             ("conv3", "bn3", planes, planes * exp, 1, 1, 0)]
                else:
                    chain = [("conv1", "bn1", inplanes, planes, 3, stride, 1), ("conv2", "bn2", planes, planes, 3, 1, 1)]
                for c, bn, ci, co, k, s_, p_ in chain:
                    self._conv(f"{pre}.{c}", ci, co, k)
                    self._bn(f"{pre}.{bn}", co)
                resid = cur
                if down:
                    self._conv(f"{pre}.downsample.0", inplanes, planes * exp, 1)
                    self._bn(f"{pre}.downsample.1", planes * exp)
                    resid = f"{pre}.ds"
                    ops.append(dict(kind=capi.BB_CONV, key=f"{pre}.downsample.0", bn=f"{pre}.downsample.1", cin=inplanes,
                                    cout=planes * exp, k=1, stride=stride, pad=0, relu=False, src=cur, dst=resid))
                x = cur
                for n, (c, bn, ci, co, k, s_, p_) in enumerate(chain):
                    last = n == len(chain) - 1
                    dst = f"{pre}.{c}"
                    ops.append(dict(kind=capi.BB_CONV, key=f"{pre}.{c}", bn=f"{pre}.{bn}", cin=ci, cout=co, k=k, stride=s_,
                                    pad=p_, relu=True, src=x, dst=dst, res=resid if last else None))
                    x = dst
                cur, inplanes = x, planes * exp
        for d, f in enumerate(self.deconv_filters):
            key, bn = f"deconv_layers.{3 * d}", f"deconv_layers.{3 * d + 1}"
            self._conv(key, inplanes, f, 4, transposed=True, bias=self.deconv_bias)
            self._bn(bn, f)
            ops.append(dict(kind=capi.BB_DECONV, key=key, bn=bn, cin=inplanes, cout=f, k=4, stride=2, pad=1, relu=True,
                            src=cur, dst=key, bias=self.deconv_bias))
            cur, inplanes = key, f
        self._conv("final_layer", inplanes, self.num_joints, 1, bias=True)
        ops.append(dict(kind=capi.BB_CONV, key="final_layer", bn=None, cin=inplanes, cout=self.num_joints, k=1, stride=1,
                        pad=0, relu=False, src=cur, dst=None, bias=True, heat=True))

    # ---- packing ----------------------------------------------------------------------------------------------------
    def mark_dirty(self):
        self.__dict__["_dirty"] = True

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.mark_dirty()
        return out

    def _device(self):
        return self.get("conv1.weight").device

    def _stream(self):
        dev = self._device()
        return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else None

    def _plan(self, H, W):
        """Op array + buffer shapes for an input of H x W (cached)."""
        if (H, W) in self._plans:
            return self._plans[(H, W)]
        shapes = {"x": (8, H, W)}            # stored as [H][W/2] pixel pairs of 8 channels (4 per pixel)
        arr = (capi.FvpBbOp * len(self._convs))()
        names = ["x"]
        w_off, e_off = 0, 64                   # eblob[0:64] stays zero: the DMA's zero page
        for i, o in enumerate(self._convs):
            cin_buf, h, w = shapes[o["src"]]
            if o["kind"] == capi.BB_MAXPOOL:
                oh, ow = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
                k = s_ = p_ = 0
                coutp = cin_buf
            elif o["kind"] == capi.BB_DECONV:
                oh, ow = 2 * h, 2 * w
                k, s_, p_ = 4, 2, 1
                coutp = _up(o["cout"], 64)
            else:
                k, s_, p_ = o["k"], o["stride"], o["pad"]
                oh, ow = (h + 2 * p_ - k) // s_ + 1, (w + 2 * p_ - k) // s_ + 1
                coutp = _up(o["cout"], 64)
            dst = -1
            if o.get("dst") is not None:
                shapes[o["dst"]] = (o["cout"], oh, ow)
                names.append(o["dst"])
                dst = len(names) - 1
            res = names.index(o["res"]) if o.get("res") else -1
            flags = (capi.EPI_RELU if o.get("relu") else 0) | (capi.BB_OUT_HEAT if o.get("heat") else 0) | \
                (capi.BB_STEM if o.get("stem") else 0)
            arr[i] = capi.FvpBbOp(o["kind"], names.index(o["src"]), dst, res, o["cin"], cin_buf, o["cout"], coutp,
                                  o["cout"], k, k, s_, p_, h, w, oh, ow, flags, w_off, e_off)
            if o["kind"] != capi.BB_MAXPOOL:
                taps = 28 if o.get("stem") else (4 if o["kind"] == capi.BB_DECONV else k * k)
                w_off += (4 if o["kind"] == capi.BB_DECONV else 1) * coutp * taps * cin_buf
                e_off += 2 * coutp
        plan = dict(ops=arr, names=names, shapes=shapes, w_elems=w_off, e_elems=e_off, out_hw=(oh, ow), tuned={})
        self._plans[(H, W)] = plan
        return plan

    def ensure_packed(self, plan):
        if not self._dirty and self.__dict__.get("_wblob") is not None:
            return
        dev = self._device()
        wblob = torch.zeros(plan["w_elems"], dtype=torch.bfloat16, device=dev)
        eblob = torch.zeros(plan["e_elems"], dtype=torch.float32, device=dev)
        s = self._stream()

        def ptr(t):
            return C.c_void_p(t.data_ptr()) if t is not None else None

        for i, o in enumerate(self._convs):
            if o["kind"] == capi.BB_MAXPOOL:
                continue
            w = self.get(o["key"] + ".weight").contiguous()
            b = self.get(o["key"] + ".bias") if o.get("bias") else None
            bn = [None] * 4
            if o.get("bn"):
                bn = [self.get(o["bn"] + s_) for s_ in (".weight", ".bias", ".running_mean", ".running_var")]
            rc = self.lib.fvp_bb_pack(ptr(w), ptr(b), *[ptr(t) for t in bn], BN_EPS, C.byref(plan["ops"][i]), ptr(wblob),
                                      ptr(eblob), s)
            capi.check(self.lib, rc, "fvp_bb_pack " + o["key"])
        if dev.type == "cuda":
            # one-time: the packed blobs may be consumed from other streams right away (one backbone shared by
            # the replicas of a PipelinedForward), so packing completes before anyone can see _dirty == False
            torch.cuda.current_stream(dev).synchronize()
        self.__dict__["_wblob"], self.__dict__["_eblob"] = wblob, eblob
        self.__dict__["_dirty"] = False

    # ---- forward ----------------------------------------------------------------------------------------------------
    def _run(self, x, want_cl, want_nchw):
        assert x.dim() == 4 and x.shape[1] == 3 and x.dtype == torch.float32
        N, _, H, W = x.shape
        assert H % 32 == 0 and W % 32 == 0, "image size must be a multiple of 32"
        dev = x.device
        plan = self._plan(H, W)
        self.ensure_packed(plan)
        s = self._stream()
        # activation buffers: freed buffers of the same size are reused (last use = last op reading them)
        last_use = {}
        for i, op in enumerate(plan["ops"]):
            for b in (op.src, op.res):
                if b >= 0:
                    last_use[b] = i
        pool, bufs = {}, [None] * len(plan["names"])
        c, h, w = plan["shapes"]["x"]
        bufs[0] = torch.empty((N, h, w // 2, c), dtype=torch.bfloat16, device=dev)
        capi.check(self.lib, self.lib.fvp_bb_input(C.c_void_p(x.contiguous().data_ptr()), C.c_void_p(bufs[0].data_ptr()), N, 3,
                                                   H, W, s), "fvp_bb_input")
        for i, op in enumerate(plan["ops"]):                     # allocation plan only (launch order = op order)
            if op.dst >= 0:
                c, h, w = plan["shapes"][plan["names"][op.dst]]
                key = (c, h, w)
                bufs[op.dst] = pool[key].pop() if pool.get(key) else torch.empty((N, h, w, c), dtype=torch.bfloat16, device=dev)
            for b in {op.src, op.res}:
                if b >= 0 and last_use.get(b) == i and b != 0:
                    t = bufs[b]
                    pool.setdefault((t.shape[3], t.shape[1], t.shape[2]), []).append(t)
        oh, ow = plan["out_hw"]
        J = self.num_joints
        JP = _up(J, 4)
        cl = torch.empty((N, oh * ow, JP), dtype=torch.float32, device=dev) if want_cl else None
        nchw = torch.empty((N, J, oh, ow), dtype=torch.float32, device=dev) if want_nchw else None
        arr = (C.c_void_p * len(bufs))(*[t.data_ptr() for t in bufs])
        if dev.type == "cuda" and self.autotune and N not in plan["tuned"] and not torch.cuda.is_current_stream_capturing():
            # once per (image size, batch): time every conv with each tile configuration of the large-tile kernel
            # and keep the fastest (the configurations compute identical bits; ~100 ms, activation buffers as scratch)
            ops = (capi.FvpBbOp * len(plan["ops"]))(*plan["ops"])
            rc = self.lib.fvp_bb_tune(ops, len(ops), C.c_void_p(self._wblob.data_ptr()), C.c_void_p(self._eblob.data_ptr()),
                                      arr, len(bufs), N, s)
            capi.check(self.lib, rc, "fvp_bb_tune")
            plan["tuned"][N] = ops                       # recorded only once the tuner has succeeded
        ops_run = plan["tuned"].get(N, plan["ops"])
        rc = self.lib.fvp_bb_run(ops_run, len(plan["ops"]), C.c_void_p(self._wblob.data_ptr()),
                                 C.c_void_p(self._eblob.data_ptr()), arr, len(bufs), N,
                                 C.c_void_p(cl.data_ptr()) if cl is not None else None, JP,
                                 C.c_void_p(nchw.data_ptr()) if nchw is not None else None, s)
        capi.check(self.lib, rc, "fvp_bb_run")
        return nchw, cl

    def forward(self, x):
        """[N,3,H,W] fp32 images -> [N,J,H/4,W/4] fp32 heatmaps (resnet.py:184-199)."""
        return self._run(x, False, True)[0]

    def forward_channels_last(self, x):
        """Heatmaps as [N, H/4 * W/4, JP] fp32 (JP = J rounded up to 4, padding channels zero)."""
        return self._run(x, True, False)[1]


def get(cfg):
    return PoseResNet(cfg)
