"""Host-side driver of the HIP hot path: owns geometry constants, packed weights and scratch
buffers, and issues the C-ABI calls (``include/fvp.h``) on the caller's current HIP stream.

No arithmetic happens here.  PyTorch is used for device memory, streams and the module /
state_dict plumbing only.  One ``HotPath`` is shared by the HDN and JLN modules of a
``FasterVoxelPoseNet`` so the channels-last heatmap staging is done once per batch.
"""
import ctypes as C

import numpy as np
import torch

from . import _capi as capi
from . import netspec

BN_EPS = 1e-5


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class SharedGeometry:
    """Per-sequence camera tables and the fine-grid coordinate cache of one device.  Read-only between (re)builds, so
    every replica of a ``PipelinedForward`` (one per batch in flight, each on its own HIP stream) uses ONE copy - each
    replica used to build and re-read its own 164 MB cache (Panoptic).  ``pending`` holds the completion events of
    EVERY (re)build not yet known to have finished, each recorded on the stream that issued it: a user on any stream
    waits for all of them before it reads or extends the tables (two consecutive batches on different streams may each
    introduce a new sequence: the second rebuild reads what the first is still writing).  Superseded tensors are kept
    alive (``retired``) only until every stream that ever used the tables has passed the point of retirement."""

    def __init__(self):
        self.cams = None          # [nsets, V, 24]
        self.seq_ids = {}
        self.fine_grid = None     # [nsets, V, F0*F1*F2, 2]
        self.fine_grid_key = None
        self.pending = []         # events of (re)builds that may still be running
        self.users = {}           # stream handle -> torch stream: every stream that has read the tables
        self.retired = []         # (tensor, [event per user stream at retirement])

    def retire(self, tensor):
        """Keep a superseded table alive until the launches already queued on every user stream have run."""
        self.retired = [(t, evs) for t, evs in self.retired if not all(e.query() for e in evs)]
        if tensor is None:
            return
        evs = []
        for st in self.users.values():
            e = torch.cuda.Event()
            e.record(st)
            evs.append(e)
        if evs:
            self.retired.append((tensor, evs))


class HotPath:
    def __init__(self, cfg, _lib=None):
        # `_lib` is a test seam (tests/hipemu); the product always loads libfvp_hip.so
        self._injected = _lib is not None
        self.lib = _lib if _lib is not None else capi.load()
        self.cfg = cfg
        self.device = torch.device(cfg.DEVICE)
        if not self._injected and self.device.type != "cuda":
            raise capi.FvpError(f"cfg.DEVICE={cfg.DEVICE!r}: the HIP path needs a ROCm GPU device "
                                "(spelled 'cuda:N' in PyTorch-ROCm); there is no CPU fallback")
        ds, cs, ins = cfg.DATASET, cfg.CAPTURE_SPEC, cfg.INDIVIDUAL_SPEC
        self.J = int(ds.NUM_JOINTS)
        self.JP = (self.J + 3) // 4 * 4
        self.W, self.H = int(ds.HEATMAP_SIZE[0]), int(ds.HEATMAP_SIZE[1])
        self.N = int(cs.MAX_PEOPLE)
        self.min_score = float(cs.MIN_SCORE)
        self.X, self.Y, self.Z = (int(v) for v in cs.VOXELS_PER_AXIS)
        self.C = int(ins.VOXELS_PER_AXIS[0])
        assert len(set(int(v) for v in ins.VOXELS_PER_AXIS)) == 1, "cubic individual volume expected"
        self.beta = float(cfg.NETWORK.BETA)
        self.F = int(cfg.NETWORK.NUM_CHANNEL_JOINT_FEAT)
        self.Hd = int(cfg.NETWORK.NUM_CHANNEL_JOINT_HIDDEN)
        dev = self.device

        # ---- constants, computed with the reference's own expressions (host, once) -----------
        # voxel-centre axes: linspace(-S/2, S/2, n) + centre (project_whole.py:34-40)
        def axes(size, center, nbins):
            return [(torch.linspace(-size[a] / 2, size[a] / 2, int(nbins[a])) + center[a]).to(dev).contiguous()
                    for a in range(3)]
        self.whole_axes = axes(cs.SPACE_SIZE, cs.SPACE_CENTER, cs.VOXELS_PER_AXIS)
        # ProposalLayer constants (human_detection_net.py:22-23)
        scale = torch.tensor(cs.SPACE_SIZE) / (torch.tensor(cs.VOXELS_PER_AXIS) - 1)
        bias = torch.tensor(cs.SPACE_CENTER) - torch.tensor(cs.SPACE_SIZE) / 2.0
        self.prop_sb = torch.cat([scale, bias]).float().to(dev).contiguous()
        # project_individual constants (project_individual.py:22-30)
        whole_c = torch.tensor(cs.SPACE_CENTER)
        whole_s = torch.tensor(cs.SPACE_SIZE)
        ind_s = torch.tensor(ins.SPACE_SIZE)
        cube = torch.tensor(ins.VOXELS_PER_AXIS, dtype=torch.int32)
        fine = (whole_s / ind_s * (cube - 1)).int() + 1
        ind_scale = (fine.float() - 1) / whole_s
        ind_bias = -ind_s / 2.0 / whole_s * (fine - 1) - ind_scale * (whole_c - whole_s / 2.0)
        self.fine = [int(v) for v in fine]
        self.ind_consts = torch.cat([ind_scale, ind_bias, whole_s, ind_s]).float().to(dev).contiguous()
        # fine[3], cube[3]: HOST array, passed to fvp_person_boxes by value (it becomes kernel arguments)
        self.fine_cube = (C.c_int32 * 6)(*(self.fine + [int(v) for v in cube]))
        self.fine_dev = torch.tensor(self.fine, dtype=torch.int32, device=dev)
        self.fine_host = (C.c_int32 * 3)(*self.fine)
        # per-sequence cache of the fine grid's sampling coordinates (what the reference caches in
        # project_individual.py:82-94; 164 MB for the Panoptic shape set): the fused tri-plane kernel then loads
        # 8 bytes per (voxel, view) instead of recomputing the projection.  Off above ~2 GB per camera set.
        self.cache_fine_grid = not self._injected     # (the CPU emulation of the kernels builds it too slowly for every test)
        self.fine_grid_limit_bytes = 2 << 30           # per camera set
        self.fine_grid_total_limit_bytes = 8 << 30     # all sets of this replica together (288 GB of HBM per GPU)
        self.geo = SharedGeometry()                    # cameras + coordinate cache (shared by pipeline replicas)
        self.fine_axes = axes(cs.SPACE_SIZE, cs.SPACE_CENTER, self.fine)
        # center_grid [3, C*C, 2] (project_individual.py:37-40): xy at z0, xz at y0, yz at x0
        ia = axes(ins.SPACE_SIZE, cs.SPACE_CENTER, ins.VOXELS_PER_AXIS)
        gx, gy, gz = (a.cpu() for a in ia)
        Cn = self.C
        xy = torch.stack([gx.view(Cn, 1).expand(Cn, Cn), gy.view(1, Cn).expand(Cn, Cn)], dim=2).reshape(-1, 2)
        xz = torch.stack([gx.view(Cn, 1).expand(Cn, Cn), gz.view(1, Cn).expand(Cn, Cn)], dim=2).reshape(-1, 2)
        yz = torch.stack([gy.view(Cn, 1).expand(Cn, Cn), gz.view(1, Cn).expand(Cn, Cn)], dim=2).reshape(-1, 2)
        self.center_grid = torch.stack([xy, xz, yz]).to(dev).contiguous()

        # ---- conv stacks --------------------------------------------------------------------------
        self.specs = {
            "center_net": netspec.centernet_spec(self.J, self.X, self.Y),
            "c2c_net": netspec.c2cnet_spec(self.J, self.Z),
            "conv_net": netspec.p2pnet_spec(self.J, self.J, self.C),
        }
        self.params = {k: torch.zeros(max(s.nparams, 4), device=dev) for k, s in self.specs.items()}
        self.wn_params = torch.zeros(self.F * 12 + self.Hd * self.F + 2 * self.Hd + 4, device=dev)
        self._geom = None
        self._geom_key = None
        self._frame_sets = {}
        self._heat_key = None
        self._heat_cl = None
        self._person_frame = {}
        self._scratch = {}
        self.fused_c2c = True        # C2CNet as one kernel per batch (False: generic conv interpreter)
        # HumanDetectionNet only ever reads N z-columns per frame of the whole-space cubes (human_detection_net.py:92-93):
        # by default they are projected directly (fvp_project_columns, same bits) and the [B,J,X,Y,Z] cubes are never
        # written.  True: materialise them (kept in self.last["cubes"]; the drop-in ProjectLayer always materialises).
        self.keep_hdn_cubes = False

    # ---------------------------------------------------------------------------------------------
    def stream(self):
        if self.device.type == "cuda":
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    def _call(self, name, *args):
        capi.check(self.lib, getattr(self.lib, name)(*args), name)

    def _check_tensor(self, t, what):
        if t.device != self.device and not (self.device.index is None and t.device.type == self.device.type):
            raise capi.FvpError(f"{what} lives on {t.device}, the model was built for {self.device} (cfg.DEVICE)")
        if t.dtype != torch.float32:
            raise capi.FvpError(f"{what} must be float32, got {t.dtype}")

    # ---- geometry / cameras ------------------------------------------------------------------------
    def geom(self, resize_transform):
        key = None
        if isinstance(resize_transform, torch.Tensor):
            key = (id(resize_transform), resize_transform.data_ptr(), resize_transform._version)
            self._geom_ref = resize_transform      # pin the tensor so id / address cannot be recycled
        if self._geom is None or key is None or key != self._geom_key:
            rt = np.asarray(resize_transform.detach().cpu() if isinstance(resize_transform, torch.Tensor)
                            else resize_transform, dtype=np.float32).reshape(6)
            ds = self.cfg.DATASET
            g = capi.FvpGeom()
            g.clamp_max = float(max(ds.ORI_IMAGE_SIZE[0], ds.ORI_IMAGE_SIZE[1]))
            for i in range(6):
                g.rt[i] = float(rt[i])
            g.hm_w, g.hm_h = float(self.W), float(self.H)
            g.img_w, g.img_h = float(ds.IMAGE_SIZE[0]), float(ds.IMAGE_SIZE[1])
            g.W, g.H = self.W, self.H
            g.V = 0
            g.J, g.JP = self.J, self.JP
            self._geom, self._geom_key = g, key
        return self._geom

    @staticmethod
    def _cam_row(cam):
        """Camera dict (lists or numpy) -> 24 fp32 values, converted like
        lib/utils/cameras.py:11-18 (float64 -> float32)."""
        row = np.zeros(capi.FVP_CAM_FLOATS, np.float32)
        row[0:9] = np.asarray(cam["R"], np.float64).reshape(9)
        row[9:12] = np.asarray(cam["T"], np.float64).reshape(3)
        row[12], row[13] = float(cam["fx"]), float(cam["fy"])
        row[14], row[15] = float(cam["cx"]), float(cam["cy"])
        row[16:19] = np.asarray(cam["k"], np.float64).reshape(3)
        row[19:21] = np.asarray(cam["p"], np.float64).reshape(2)
        return row

    def frame_sets(self, meta, cameras, V):
        """Per-frame camera-set ids (one set per sequence), uploading new sequences once."""
        geo = self.geo
        seqs = tuple(meta["seq"])
        new = [s for s in dict.fromkeys(seqs) if s not in geo.seq_ids]
        self._await_geometry()                          # before reading OR extending geo.cams
        for s in new:
            assert s in cameras.keys(), "missing camera parameters for the current sequence"
            assert len(cameras[s]) == V, "inconsistent number of cameras"
            rows = np.stack([self._cam_row(cameras[s][c]) for c in range(V)])
            t = torch.from_numpy(rows).to(self.device)
            geo.retire(geo.cams)
            geo.cams = t[None] if geo.cams is None else torch.cat([geo.cams, t[None]], dim=0)
            geo.seq_ids[s] = geo.cams.shape[0] - 1      # ids are append-only: other replicas' frame-set tensors stay valid
        if new:
            self._geometry_changed()
        if seqs not in self._frame_sets:
            if len(self._frame_sets) >= 256:            # bounded: a long run over many sequence mixes
                self._frame_sets.pop(next(iter(self._frame_sets)))
            self._frame_sets[seqs] = torch.tensor([geo.seq_ids[s] for s in seqs], dtype=torch.int32,
                                                  device=self.device)
        return self._frame_sets[seqs]

    def _geometry_changed(self):
        """Mark the shared tables as (re)built by work queued on the current stream."""
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.geo.pending.append(ev)

    def _await_geometry(self):
        """Order the current stream behind EVERY (re)build of the shared tables that may still be running (a no-op once
        they have completed) and register the stream as a user of the tables."""
        if self.device.type != "cuda":
            return
        st = torch.cuda.current_stream(self.device)
        geo = self.geo
        geo.users.setdefault(st.cuda_stream, st)
        if geo.pending:
            geo.pending = [ev for ev in geo.pending if not ev.query()]
            for ev in geo.pending:
                st.wait_event(ev)

    def cams_of(self, seq):
        return self.geo.cams[self.geo.seq_ids[seq]]

    def fine_grid_cache(self, resize_transform, V):
        """[nsets, V, F0*F1*F2, 2] sampling coordinates of the fine grid for every uploaded camera set (built by
        fvp_sample_grid, extended when a new sequence appears), or None when one set would exceed
        ``fine_grid_limit_bytes`` or all sets together ``fine_grid_total_limit_bytes`` (the kernel then recomputes
        the projection: same bits).  The cache belongs to one (resize_transform, V): it is rebuilt when
        ``geom()`` re-derives the geometry, never silently reused."""
        geo = self.geo
        n = self.fine[0] * self.fine[1] * self.fine[2]
        nsets = geo.cams.shape[0]
        if V * n * 8 > self.fine_grid_limit_bytes or nsets * V * n * 8 > self.fine_grid_total_limit_bytes:
            geo.retire(geo.fine_grid)
            geo.fine_grid = None
            return None
        g = self.geom(resize_transform)
        key = (tuple(g.rt), g.clamp_max, V)
        if geo.fine_grid is not None and geo.fine_grid_key != key:
            geo.retire(geo.fine_grid)
            geo.fine_grid = None                        # resize_transform changed: stale coordinates
        have = 0 if geo.fine_grid is None else geo.fine_grid.shape[0]
        if have < nsets:
            g.V = V
            self._await_geometry()                      # (another replica may still be writing the part that is copied)
            new = torch.empty((nsets, V, n, 2), device=self.device)
            if have:
                new[:have] = geo.fine_grid
                geo.retire(geo.fine_grid)
            fa = self.fine_axes
            for i in range(have, nsets):
                self._call("fvp_sample_grid", _ptr(fa[0]), _ptr(fa[1]), _ptr(fa[2]), self.fine[0], self.fine[1],
                           self.fine[2], _ptr(geo.cams[i]), C.byref(g), _ptr(new[i]), self.stream())
            geo.fine_grid, geo.fine_grid_key = new, key
            self._geometry_changed()
        else:
            self._await_geometry()
        return geo.fine_grid

    # ---- staging ---------------------------------------------------------------------------------------
    def heat_cl(self, heatmaps, g, reuse=False):
        """Channels-last staging of one batch.  ``reuse=True`` (only passed by
        FasterVoxelPoseNet.forward, which hands the very same tensor to HDN and JLN) skips the
        restaging when the tensor identity matches; any other caller restages."""
        self._check_tensor(heatmaps, "heatmaps")
        heatmaps = heatmaps.contiguous()
        key = (heatmaps.data_ptr(), heatmaps._version, tuple(heatmaps.shape))
        if not (reuse or self.__dict__.get("_adopted")) or key != self._heat_key:
            self._adopted = False
            B, V = heatmaps.shape[:2]
            out = self.scratch("heat_cl", (B, V, self.H, self.W, self.JP))
            self._call("fvp_heatmaps_to_cl", _ptr(heatmaps), _ptr(out), B, C.byref(g), self.stream())
            self._heat_cl, self._heat_key = out, key
        return self._heat_cl

    def adopt_staging(self, heatmaps, heat_cl):
        """The producer of the heatmaps (the bf16 backbone) already wrote the channels-last copy
        [B,V,H*W,JP]: use it for the next forward over exactly this ``heatmaps`` tensor."""
        B, V = heatmaps.shape[:2]
        assert heat_cl.is_contiguous() and heat_cl.numel() == B * V * self.H * self.W * self.JP
        self._heat_cl = heat_cl.view(B, V, self.H, self.W, self.JP)
        self._heat_key = (heatmaps.data_ptr(), heatmaps._version, tuple(heatmaps.shape))
        self._adopted = True

    def invalidate_staging(self):
        self._heat_key = None
        self._adopted = False

    def scratch(self, name, shape, dtype=torch.float32, zero=False):
        key = (name, tuple(shape), dtype)
        t = self._scratch.get(key)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._scratch[key] = t
        if zero:
            t.zero_()
        return t

    # ---- weights -----------------------------------------------------------------------------------------
    def pack_stack(self, name, tree):
        """state_dict tensors of one conv stack -> packed blob (fvp_pack_conv per conv)."""
        spec, blob = self.specs[name], self.params[name]
        s = self.stream()
        for key, bn, transposed, oi in spec.param_keys:
            w = tree.get(key + ".weight")
            self._check_tensor(w, key)
            b = tree.get(key + ".bias")
            bnp = [None] * 4
            if bn is not None:
                bnp = [tree.get(bn + ".weight"), tree.get(bn + ".bias"), tree.get(bn + ".running_mean"),
                       tree.get(bn + ".running_var")]
            self._call("fvp_pack_conv", _ptr(w.contiguous()), _ptr(b), *[_ptr(t) for t in bnp], BN_EPS,
                       1 if transposed else 0, C.byref(spec.op_array[oi]), _ptr(blob), s)

    def pack_weightnet(self, tree):
        g = tree.get
        self._call("fvp_pack_weightnet", _ptr(g("heatmap_feature_net.0.weight").contiguous()),
                   _ptr(g("heatmap_feature_net.0.bias")), _ptr(g("heatmap_feature_net.1.weight")),
                   _ptr(g("heatmap_feature_net.1.bias")), _ptr(g("heatmap_feature_net.1.running_mean")),
                   _ptr(g("heatmap_feature_net.1.running_var")), BN_EPS, _ptr(g("output.0.weight").contiguous()),
                   _ptr(g("output.0.bias")), _ptr(g("output.2.weight").contiguous()), _ptr(g("output.2.bias")),
                   self.F, self.Hd, _ptr(self.wn_params), self.stream())

    # ---- conv stack ----------------------------------------------------------------------------------------
    def run_stack(self, name, x, planes, plane_valid=None, valid_div=1, fresh=()):
        """`fresh`: output names whose buffer is a new tensor instead of the stack's reused scratch - the caller may hand it
        out (HDN's public heatmaps) without a copy launch."""
        spec = self.specs[name]
        bufs = [x]
        own = {spec.outputs[k] for k in fresh}
        for i, (c, h, w) in enumerate(spec.bufs[1:], start=1):
            bufs.append(torch.empty((planes, c, h, w), device=self.device) if i in own
                        else self.scratch(f"{name}.buf{i}", (planes, c, h, w)))
        arr = (C.c_void_p * len(bufs))(*[t.data_ptr() for t in bufs])
        self._call("fvp_conv_stack_run", spec.op_array, len(spec.ops), _ptr(self.params[name]), arr, len(bufs),
                   planes, _ptr(plane_valid), valid_div, self.stream())
        return {k: bufs[i] for k, i in spec.outputs.items()}

    def run_stack_fused_1d(self, name, x, planes, fresh=False):
        """Whole 1-D stack in one kernel (fvp_conv_stack_run_fused_1d); x = [planes, cin, 1, W]."""
        spec = self.specs[name]
        c, h, w = spec.bufs[spec.outputs["out"]]
        out = torch.empty((planes, c, h, w), device=self.device) if fresh else self.scratch(f"{name}.fused_out", (planes, c, h, w))
        self._call("fvp_conv_stack_run_fused_1d", spec.op_array, len(spec.ops), _ptr(self.params[name]), _ptr(x),
                   _ptr(out), planes, self.stream())
        return {"out": out}

    # ---- operator groups ------------------------------------------------------------------------------------
    def project_whole(self, heatmaps, meta, cameras, resize_transform, want_cubes=True, want_zmax=False):
        B, V = heatmaps.shape[:2]
        g = self.geom(resize_transform)
        g.V = V
        fs = self.frame_sets(meta, cameras, V)
        hcl = self.heat_cl(heatmaps, g)
        cubes = torch.empty((B, self.J, self.X, self.Y, self.Z), device=self.device) if want_cubes else None
        zmax = self.scratch("zmax", (B, self.J, self.X, self.Y)) if want_zmax else None
        ax = self.whole_axes
        self._call("fvp_project_whole", _ptr(hcl), _ptr(self.geo.cams), _ptr(fs), _ptr(ax[0]), _ptr(ax[1]), _ptr(ax[2]),
                   self.X, self.Y, self.Z, B, C.byref(g), _ptr(cubes), _ptr(zmax), self.stream())
        return cubes, zmax

    def sample_grid(self, axes3, seq, resize_transform, V):
        g = self.geom(resize_transform)
        g.V = V
        n = axes3[0].numel() * axes3[1].numel() * axes3[2].numel()
        grid = torch.empty((V, 1, n, 2), device=self.device)
        self._call("fvp_sample_grid", _ptr(axes3[0]), _ptr(axes3[1]), _ptr(axes3[2]), axes3[0].numel(),
                   axes3[1].numel(), axes3[2].numel(), _ptr(self.cams_of(seq)), C.byref(g), _ptr(grid), self.stream())
        return grid

    def hdn(self, heatmaps, meta, cameras, resize_transform):
        """HumanDetectionNet.forward (human_detection_net.py:76-104), inference branch."""
        B = heatmaps.shape[0]
        N, J, X, Y, Z = self.N, self.J, self.X, self.Y, self.Z
        s = self.stream()
        keep = self.keep_hdn_cubes
        cubes, zmax = self.project_whole(heatmaps, meta, cameras, resize_transform, keep, True)
        heads = self.run_stack("center_net", zmax, B, fresh=("output_hm",))     # returned to the caller: its own tensor
        hm2d = heads["output_hm"]
        bbox_map = heads["output_size"]
        dev = self.device
        conf2d = self.scratch("conf2d", (B, N))
        idx2d = self.scratch("idx2d", (B, N, 2), torch.int64)
        flat = self.scratch("flat", (B, N), torch.int64)
        self._call("fvp_nms_topk", _ptr(hm2d), B, X, Y, N, _ptr(conf2d), _ptr(idx2d), _ptr(flat), s)
        bbox_flat = torch.empty((B, X * Y, 2), device=dev)
        match_bbox = self.scratch("match_bbox", (B, N, 2))
        feat1d = self.scratch("feat1d", (B * N, J, 1, Z))
        if keep:
            self._call("fvp_gather_proposals", _ptr(bbox_map), _ptr(cubes), _ptr(flat), B, J, X, Y, Z, N,
                       _ptr(bbox_flat), _ptr(match_bbox), _ptr(feat1d), s)
        else:
            self._call("fvp_gather_proposals", _ptr(bbox_map), None, _ptr(flat), B, J, X, Y, Z, N, _ptr(bbox_flat),
                       _ptr(match_bbox), None, s)
            g = self.geom(resize_transform)
            ax = self.whole_axes
            self._call("fvp_project_columns", _ptr(self._heat_cl), _ptr(self.geo.cams),
                       _ptr(self.frame_sets(meta, cameras, heatmaps.shape[1])), _ptr(ax[0]), _ptr(ax[1]), _ptr(ax[2]),
                       X, Y, Z, B, C.byref(g), _ptr(flat), N, _ptr(feat1d), s)
        if self.fused_c2c and Z <= 24:
            hm1d = self.run_stack_fused_1d("c2c_net", feat1d, B * N, fresh=True)["out"].view(B, N, Z)
        else:
            hm1d = self.run_stack("c2c_net", feat1d, B * N, fresh=("out",))["out"].view(B, N, Z)
        centers = torch.empty((B, N, 7), device=dev)
        topk_index = self.scratch("topk_index", (B, N, 3), torch.int64)
        # `mask = proposal_centers[:, :, 3] >= 0` (faster_voxelpose.py:45) comes out of the same launch (ABI 8)
        valid = self.scratch("proposal_valid", (B, N), torch.uint8)
        self._call("fvp_proposals", _ptr(hm1d), _ptr(conf2d), _ptr(idx2d), _ptr(match_bbox), _ptr(self.prop_sb),
                   self.min_score, B, N, Z, _ptr(topk_index), _ptr(centers), _ptr(valid), s)
        self.last = dict(cubes=cubes, zmax=zmax, conf2d=conf2d, idx2d=idx2d, flat=flat, topk_index=topk_index,
                         bbox_map=bbox_map, feat1d=feat1d, hm2d=hm2d, hm1d=hm1d, bbox_flat=bbox_flat, valid=valid)
        return hm2d, hm1d, centers, bbox_flat

    def proposal_layer(self, topk_index, topk_confs, match_bbox, min_score):
        """ProposalLayer.forward as a standalone launch (human_detection_net.py:44-65, eval branch)."""
        self._check_tensor(topk_confs, "topk_confs")
        self._check_tensor(match_bbox, "match_bbox_preds")
        B, N = topk_confs.shape
        idx = topk_index.to(torch.int64).contiguous()
        assert idx.shape == (B, N, 3) and match_bbox.shape == (B, N, 2)
        centers = torch.empty((B, N, 7), device=self.device)
        self._call("fvp_proposal_layer", _ptr(idx), _ptr(topk_confs.contiguous()), _ptr(match_bbox.contiguous()),
                   _ptr(self.prop_sb), float(min_score), B, N, _ptr(centers), self.stream())
        return centers

    def person_frame(self, B, N):
        key = (B, N)
        if key not in self._person_frame:
            self._person_frame[key] = (torch.arange(B * N, device=self.device) // N).to(torch.int32)
        return self._person_frame[key]

    def person_boxes(self, centers2d):
        n = centers2d.shape[0]
        boxes = self.scratch("boxes", (n, 9), torch.int32)
        offset = self.scratch("offset", (n, 3))
        self._call("fvp_person_boxes", _ptr(centers2d), n, _ptr(self.ind_consts), self.fine_cube, _ptr(boxes),
                   _ptr(offset), self.stream())
        return boxes, offset

    def softargmax_weightnet(self, joint_features, grids=None):
        """Standalone launch of the fused soft-argmax + WeightNet kernel on features in the
        reference's layout [3, P, J, C, C] (joint_localization_net.py:20-34, weight_net.py:69-80).
        Returns pose [3,P,J,2], confs [P], weights [3P,J,1]."""
        x = joint_features
        self._check_tensor(x, "joint_features")
        assert x.dim() == 5 and x.shape[0] == 3 and x.shape[2] == self.J and x.shape[3] == x.shape[4] == self.C
        P, J, Cn = x.shape[1], self.J, self.C
        s = self.stream()
        feat = x.permute(1, 0, 2, 3, 4).contiguous()                # kernel layout [P][3][J][C*C] (a copy, no arithmetic)
        grid = self.center_grid if grids is None else grids.reshape(3, Cn * Cn, 2).contiguous()
        pose2d = torch.empty((P, 3, J, 2), device=self.device)
        pmax = torch.empty((P, 3, J), device=self.device)
        wgt = torch.empty((P, 3, J), device=self.device)
        if P == 0:
            return pose2d.permute(1, 0, 2, 3), torch.empty((0,), device=self.device), wgt.reshape(0, J, 1)
        self._call("fvp_softargmax_weightnet", _ptr(feat), _ptr(grid), _ptr(self.wn_params), self.beta, P, J, Cn,
                   self.F, self.Hd, None, _ptr(pose2d), _ptr(pmax), _ptr(wgt), s)
        # confs = mean over (plane, joint) of the softmax maxima: the fusion kernel computes it
        centers = torch.zeros((P, 7), device=self.device)
        offset = torch.zeros((P, 3), device=self.device)
        fused5 = torch.empty((P, J, 5), device=self.device)
        planes = torch.empty((3, P, J, 2), device=self.device)
        self._call("fvp_fuse_poses", _ptr(pose2d), _ptr(pmax), _ptr(wgt), _ptr(offset), None, P, J, _ptr(centers),
                   _ptr(fused5), _ptr(planes), s)
        return pose2d.permute(1, 0, 2, 3).contiguous(), centers[:, 4].clone(), \
            wgt.permute(1, 0, 2).reshape(3 * P, J, 1).contiguous()

    def jln(self, meta, heatmaps, proposal_centers, mask, cameras, resize_transform, fused=True,
            reuse_staging=False):
        """JointLocalizationNet.forward (joint_localization_net.py:64-99) for all B*N proposal
        slots at once; invalid slots are skipped inside the kernels.  Writes the JLN
        confidence into proposal_centers[..., 4] in place, as the reference does (:98)."""
        B, N = proposal_centers.shape[:2]
        V = heatmaps.shape[1]
        J, Cn = self.J, self.C
        nP = B * N
        s = self.stream()
        self._check_tensor(proposal_centers, "proposal_centers")
        assert proposal_centers.is_contiguous(), "proposal_centers must be contiguous (it is updated in place)"
        g = self.geom(resize_transform)
        g.V = V
        fs = self.frame_sets(meta, cameras, V)
        hcl = self.heat_cl(heatmaps, g, reuse=reuse_staging)
        # (the forward hands over the uint8 flags fvp_proposals wrote: no launch; a caller's bool mask is converted)
        valid = mask.reshape(-1) if (mask.dtype == torch.uint8 and mask.is_contiguous()) else mask.reshape(-1).to(torch.uint8).contiguous()
        pf = self.person_frame(B, N)
        centers2d = proposal_centers.view(nP, 7)
        boxes, offset = self.person_boxes(centers2d)
        fa = self.fine_axes
        planes = self.scratch("planes", (nP, 3, J, Cn, Cn), zero=True)
        if fused:
            fgrid = self.fine_grid_cache(resize_transform, V) if self.cache_fine_grid else None
            self._call("fvp_project_individual_triplane", _ptr(hcl), _ptr(self.geo.cams), _ptr(fs), _ptr(pf), _ptr(valid),
                       _ptr(boxes), _ptr(fa[0]), _ptr(fa[1]), _ptr(fa[2]), self.fine_host, Cn, nP, C.byref(g),
                       _ptr(planes), N, _ptr(fgrid), s)
        else:
            cubes = self.scratch("person_cubes", (nP, J, Cn, Cn, Cn))
            self._call("fvp_project_individual", _ptr(hcl), _ptr(self.geo.cams), _ptr(fs), _ptr(pf), _ptr(valid),
                       _ptr(boxes), _ptr(fa[0]), _ptr(fa[1]), _ptr(fa[2]), _ptr(self.fine_dev), Cn, nP, C.byref(g),
                       _ptr(cubes), s)
            self._call("fvp_triplane_max", _ptr(cubes), _ptr(planes), nP, J, Cn, s)
        feat = self.run_stack("conv_net", planes.view(nP * 3, J, Cn, Cn), nP * 3, valid, 3)["out"]
        pose2d = self.scratch("pose2d", (nP, 3, J, 2))
        pmax = self.scratch("pmax", (nP, 3, J))
        wgt = self.scratch("wgt", (nP, 3, J))
        self._call("fvp_softargmax_weightnet", _ptr(feat), _ptr(self.center_grid), _ptr(self.wn_params), self.beta, nP,
                   J, Cn, self.F, self.Hd, _ptr(valid), _ptr(pose2d), _ptr(pmax), _ptr(wgt), s)
        fused5 = torch.empty((B, N, J, 5), device=self.device)
        plane_poses = torch.empty((3, B, N, J, 2), device=self.device)
        self._call("fvp_fuse_poses", _ptr(pose2d), _ptr(pmax), _ptr(wgt), _ptr(offset), _ptr(valid), nP, J,
                   _ptr(centers2d), _ptr(fused5), _ptr(plane_poses), s)
        self.last_jln = dict(planes=planes, feat=feat, boxes=boxes, offset=offset, pose2d=pose2d, pmax=pmax, wgt=wgt,
                             valid=valid)
        return fused5, plane_poses
