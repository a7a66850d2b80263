"""Declarative description of the three conv stacks of the hot path.

From one description of the topology this module derives
  * the parameter tree with exactly the reference's ``state_dict`` key names
    (lib/models/cnns_2d.py:115-187, lib/models/cnns_1d.py:112-143), and
  * the ``FvpConvOp`` list the HIP interpreter (``fvp_conv_stack_run``) executes, with the
    packed-parameter offsets.

Nothing here does arithmetic; BatchNorm/bias/ReLU/residual are epilogue flags of the conv
kernel, pooling and transposed conv are ops of their own.
"""
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _capi as capi


def _round_up(x, m):
    return (x + m - 1) // m * m


class ParamTree(nn.Module):
    """Generic container reproducing arbitrary dotted state_dict keys (children may be
    named '0', '1', ... like nn.Sequential's)."""

    def add(self, dotted, tensor, buffer=False):
        node = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, ParamTree())
            node = node._modules[p]
        if buffer:
            node.register_buffer(parts[-1], tensor)
        else:
            node.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))

    def get(self, dotted):
        node = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            node = node._modules[p]
        leaf = parts[-1]
        return node._parameters[leaf] if leaf in node._parameters else node._buffers[leaf]

    def has(self, dotted):
        node = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if p not in node._modules:
                return False
            node = node._modules[p]
        return parts[-1] in node._parameters or parts[-1] in node._buffers

    def forward(self, *a, **k):  # pragma: no cover - containers are never called
        raise RuntimeError("ParamTree holds parameters only")


# Rows that do not divide the Winograd workgroup tile (CenterNet's 80 / 40 / 20-wide levels) run with masked tiles when
# they are at least WINO_MASKED_MIN_W columns wide (round 6; kWinoMaskedMinW in csrc/fvp_conv_wino.hip), narrower ones on the
# direct kernel.  WINO_GENERIC: set by tests that load the diagnostics build with FVP_WINO_GENERIC=1 (every width);
# nothing here reads the environment.
WINO_MASKED_MIN_W = 40
WINO_GENERIC = False


def winograd_shape(kh, kw, h, w, cinp, coutp, cin=None, cout=None):
    """3x3 layers the F(2x2,3x3) kernel covers (the same rule as wino_tiling() in
    csrc/fvp_conv.hip): decided from the layer shape alone, never from the batch."""
    if (kh, kw) != (3, 3) or h < 2 or h % 2 or w < 8 or w % 4 or (coutp != 32 and coutp % 64) or cinp % 4:
        return False
    if cin is not None and cin != cinp:          # whole channel chunks only (no padded input channels)
        return False
    if cout is not None and (cout % 32 or cout != coutp):   # whole 32-cout blocks, no padded couts (the epilogue has no per-cout predicate)
        return False
    if w & (w - 1) and w < WINO_MASKED_MIN_W and not WINO_GENERIC:
        return False                             # narrow rows that do not divide the workgroup tile: direct kernel
    return w // 2 <= (128 if coutp == 32 else 64)   # a tile row fits the workgroup's 16 * WT tiles


class StackSpec:
    """Op list + parameter keys of one conv stack (dim = 1 or 2)."""

    def __init__(self, dim, cin, hw):
        self.dim = dim
        self.ops = []                 # dicts, turned into FvpConvOp by finalize()
        self.bufs = [(cin,) + tuple(hw)]   # (C, H, W) per activation buffer; 0 = input
        self.param_keys = []          # (conv key, bn key or None, transposed, op index)
        self.entries = OrderedDict()  # state_dict key -> (shape, is_buffer, init)
        self.outputs = {}
        self.nparams = 0

    # -- parameters (registration order = reference module order) ---------------------------
    def _conv_entries(self, key, cin, cout, k, transposed=False):
        ks = (k,) * self.dim
        shape = ((cin, cout) if transposed else (cout, cin)) + ks
        self.entries[key + ".weight"] = (shape, False)
        self.entries[key + ".bias"] = ((cout,), False)

    def _bn_entries(self, key, c):
        self.entries[key + ".weight"] = ((c,), False)
        self.entries[key + ".bias"] = ((c,), False)
        self.entries[key + ".running_mean"] = ((c,), True)
        self.entries[key + ".running_var"] = ((c,), True)
        self.entries[key + ".num_batches_tracked"] = ((), True)

    def declare_basic(self, pre, cin, cout, k):
        self._conv_entries(pre + ".block.0", cin, cout, k)
        self._bn_entries(pre + ".block.1", cout)

    def declare_res(self, pre, cin, cout):
        self._conv_entries(pre + ".res_branch.0", cin, cout, 3)
        self._bn_entries(pre + ".res_branch.1", cout)
        self._conv_entries(pre + ".res_branch.3", cout, cout, 3)
        self._bn_entries(pre + ".res_branch.4", cout)
        if cin != cout:
            self._conv_entries(pre + ".skip_con.0", cin, cout, 1)
            self._bn_entries(pre + ".skip_con.1", cout)

    def declare_up(self, pre, cin, cout):
        self._conv_entries(pre + ".block.0", cin, cout, 2, transposed=True)
        self._bn_entries(pre + ".block.1", cout)

    # -- ops ------------------------------------------------------------------------------------
    def _new_buf(self, c, h, w):
        self.bufs.append((c, h, w))
        return len(self.bufs) - 1

    def conv(self, key, bn, src, cout, k, relu, res=None, res_after_relu=False):
        cin, h, w = self.bufs[src]
        dst = self._new_buf(cout, h, w)
        flags = (capi.EPI_RELU if relu else 0) | (capi.EPI_RES if res is not None else 0) | \
            (capi.EPI_RES_AFTER_RELU if res_after_relu else 0)
        kh = k if self.dim == 2 else 1
        self.ops.append(dict(kind=capi.OP_CONV, src=src, dst=dst, res=-1 if res is None else res, cin=cin,
                             cout=cout, kh=kh, kw=k, h=h, w=w, flags=flags))
        self.param_keys.append((key, bn, False, len(self.ops) - 1))
        return dst

    def pool(self, src):
        c, h, w = self.bufs[src]
        dst = self._new_buf(c, h // 2 if self.dim == 2 else 1, w // 2)
        self.ops.append(dict(kind=capi.OP_POOL2, src=src, dst=dst, res=-1, cin=c, cout=c, kh=1, kw=1, h=h, w=w,
                             flags=0))
        return dst

    def up(self, key, bn, src, cout, skip):
        cin, h, w = self.bufs[src]
        dst = self._new_buf(cout, h * 2 if self.dim == 2 else 1, w * 2)
        flags = capi.EPI_RELU | capi.EPI_RES | capi.EPI_RES_AFTER_RELU
        self.ops.append(dict(kind=capi.OP_CONVT2, src=src, dst=dst, res=skip, cin=cin, cout=cout,
                             kh=2 if self.dim == 2 else 1, kw=2, h=h, w=w, flags=flags))
        self.param_keys.append((key, bn, True, len(self.ops) - 1))
        return dst

    # -- reference blocks -------------------------------------------------------------------------
    def basic(self, pre, x, cout, k):
        return self.conv(pre + ".block.0", pre + ".block.1", x, cout, k, relu=True)

    def res(self, pre, x, cout):
        cin = self.bufs[x][0]
        h = self.conv(pre + ".res_branch.0", pre + ".res_branch.1", x, cout, 3, relu=True)
        s = x if cin == cout else self.conv(pre + ".skip_con.0", pre + ".skip_con.1", x, cout, 1, relu=False)
        return self.conv(pre + ".res_branch.3", pre + ".res_branch.4", h, cout, 3, relu=True, res=s)

    def trunk(self, cin):
        """front_layers + EncoderDecorder (cnns_2d.py:74-112,122-127); returns the 32-ch buffer."""
        ed = "encoder_decoder"
        # parameter registration order of the reference modules
        self.declare_basic("front_layers.0", cin, 16, 7)
        self.declare_res("front_layers.1", 16, 32)
        self.declare_res(ed + ".encoder_res1", 32, 64)
        self.declare_res(ed + ".encoder_res2", 64, 128)
        self.declare_res(ed + ".mid_res", 128, 128)
        self.declare_res(ed + ".decoder_res2", 128, 128)
        self.declare_up(ed + ".decoder_upsample2", 128, 64)
        self.declare_res(ed + ".decoder_res1", 64, 64)
        self.declare_up(ed + ".decoder_upsample1", 64, 32)
        self.declare_res(ed + ".skip_res1", 32, 32)
        self.declare_res(ed + ".skip_res2", 64, 64)
        # execution order (EncoderDecorder.forward, cnns_2d.py:91-112)
        x = self.basic("front_layers.0", 0, 16, 7)
        x = self.res("front_layers.1", x, 32)
        skip1 = self.res(ed + ".skip_res1", x, 32)
        x = self.res(ed + ".encoder_res1", self.pool(x), 64)
        skip2 = self.res(ed + ".skip_res2", x, 64)
        x = self.res(ed + ".encoder_res2", self.pool(x), 128)
        x = self.res(ed + ".mid_res", x, 128)
        x = self.res(ed + ".decoder_res2", x, 128)
        x = self.up(ed + ".decoder_upsample2.block.0", ed + ".decoder_upsample2.block.1", x, 64, skip2)
        x = self.res(ed + ".decoder_res1", x, 64)
        x = self.up(ed + ".decoder_upsample1.block.0", ed + ".decoder_upsample1.block.1", x, 32, skip1)
        return x

    def finalize(self):
        """Assign packed-parameter offsets and build the ctypes op array.  The first 64 floats of
        the blob stay zero: the conv kernel's LDS-DMA reads them for padding rows."""
        off = 64
        arr = (capi.FvpConvOp * len(self.ops))()
        for i, o in enumerate(self.ops):
            cinp = _round_up(o["cin"], 2)
            coutp = _round_up(o["cout"], 32)
            w_off = e_off = wino_off = pair_off = 0
            if o["kind"] != capi.OP_POOL2:
                w_off = off
                off += _round_up(cinp * o["kh"] * o["kw"] * coutp, 4)
                e_off = off
                off += _round_up(3 * coutp, 4)
                if o["kind"] == capi.OP_CONV and winograd_shape(o["kh"], o["kw"], o["h"], o["w"], cinp, coutp, o["cin"], o["cout"]):
                    wino_off = off
                    off += cinp * coutp * 16
                if (o["kind"] == capi.OP_CONV and (o["kh"], o["kw"]) == (7, 7) and o["cout"] <= 16
                        and o["w"] % 4 == 0):
                    pair_off = off                      # pixel-pair layout [cinp][7][8][32] ...
                    off += cinp * 7 * 8 * 32
                    ncg = 4 if o["cin"] <= 16 else -(-o["cin"] // 4)
                    off += ncg * 13 * 64 * 4            # ... + the k-grouped copy [max(4, ceil(cin/4))][13 tap quads][4][16][4] (k_conv7)
                if (o["kind"] == capi.OP_CONVT2 and o["h"] > 1 and o["w"] % 4 == 0 and coutp in (32, 64)):
                    pair_off = off                      # column-tap pairs [dy][cinp][dx*coutp + co]
                    off += 4 * cinp * coutp
            arr[i] = capi.FvpConvOp(o["kind"], o["src"], o["dst"], o["res"], o["cin"], o["cout"], o["kh"], o["kw"],
                                    o["h"], o["w"], o["flags"], w_off, e_off, cinp, coutp, wino_off, pair_off)
        self.nparams = off
        self.op_array = arr
        return self

    def build_tree(self, tree=None, prefix=""):
        """Create zero-initialised parameters/buffers with the reference's key names."""
        tree = tree if tree is not None else ParamTree()
        for key, (shape, is_buf) in self.entries.items():
            if key.endswith("num_batches_tracked"):
                t = torch.zeros((), dtype=torch.long)
            elif key.endswith("running_var"):
                t = torch.ones(shape)
            else:
                t = torch.zeros(shape)
            tree.add(prefix + key, t, buffer=is_buf)
        return tree


def centernet_spec(cin, X, Y):
    """CenterNet (cnns_2d.py:147-178); input = z-max map [B,cin,X,Y]."""
    s = StackSpec(2, cin, (X, Y))
    x = s.trunk(cin)
    for name, c in (("output_hm", 1), ("output_size", 2)):
        s._conv_entries(name + ".0", 32, 32, 3)
        s._conv_entries(name + ".2", 32, c, 1)
    for name, c in (("output_hm", 1), ("output_size", 2)):
        h = s.conv(name + ".0", None, x, 32, 3, relu=True)
        s.outputs[name] = s.conv(name + ".2", None, h, c, 1, relu=False)
    return s.finalize()


def p2pnet_spec(cin, cout, C):
    """P2PNet (cnns_2d.py:115-135)."""
    s = StackSpec(2, cin, (C, C))
    x = s.trunk(cin)
    s._conv_entries("output_layer", 32, cout, 1)
    s.outputs["out"] = s.conv("output_layer", None, x, cout, 1, relu=False)
    return s.finalize()


def c2cnet_spec(cin, Z):
    """C2CNet (cnns_1d.py:112-132): the same topology in 1-D (H = 1)."""
    s = StackSpec(1, cin, (1, Z))
    x = s.trunk(cin)
    s._conv_entries("output_hm", 32, 1, 1)
    s.outputs["out"] = s.conv("output_hm", None, x, 1, 1, relu=False)
    return s.finalize()
