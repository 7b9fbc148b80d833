"""ctypes binding of libcistaflow.so (include/cistaflow.h).

The product path has no CPU fallback: if the HIP library is missing or cannot be loaded this
module raises, and every op that needs it raises with it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CF_LIB_PATH: tuning builds of the same library (tools/); the product always loads the in-tree file
LIB_PATH = os.environ.get("CF_LIB_PATH") or os.path.join(_HERE, "libcistaflow.so")

CF_MODE_CISTA, CF_MODE_EIFLOW, CF_MODE_ERAFT, CF_MODE_IDNET = 0, 1, 2, 3
CF_WARP_FORWARD, CF_WARP_BACKWARD = 0, 1
PRECISIONS = {"f32": 0, "f16x3": 3, "f16": 1}

# every symbol include/cistaflow.h declares (tests check the .so exports exactly these)
SYMBOLS = [
    "cf_create", "cf_destroy", "cf_last_error", "cf_workspace_bytes", "cf_load_weights",
    "cf_finalize_weights", "cf_warp", "cf_cista_forward", "cf_flow_forward", "cf_step",
    "cf_op_conv2d", "cf_op_instance_norm_relu", "cf_op_corr_lookup", "cf_op_nchw_to_nhwc",
    "cf_op_nhwc_to_nchw", "cf_profile_enable", "cf_profile_read", "cf_conv_tile_name", "cf_profile_report", "cf_op_conv2d_bench", "cf_events_to_voxel", "cf_op_conv2d_inorm_stats", "cf_quantize_u8", "cf_hint_prev_grid",
    "cf_profile_report_json", "cf_metrics_scratch_doubles", "cf_metrics_recon", "cf_metrics_flow", "cf_metrics_fwl",
    "cf_graph_enable", "cf_graph_stats", "cf_events_to_voxel_ex", "cf_voxel_preprocess", "cf_metrics_ssim",
    "cf_conv_tile_mfma_ratio", "cf_plan_enable", "cf_plan_json", "cf_conv_plan", "cf_flow_to_bgr",
]


class cf_config(C.Structure):
    _fields_ = [
        ("mode", C.c_int), ("batch", C.c_int), ("height", C.c_int), ("width", C.c_int),
        ("num_bins", C.c_int), ("base_channels", C.c_int), ("depth", C.c_int), ("iters", C.c_int),
        ("warp_mode", C.c_int), ("device", C.c_int), ("precision", C.c_int),
    ]


_lib = None


def load():
    """Load libcistaflow.so (built by __graft_entry__.build()); raises if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libcistaflow.so not found at %s -- build it first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback for the CISTA-Flow hot path." % LIB_PATH)
    # torch first: PyTorch-ROCm bundles its own libamdhip64.so.7; loading it before ours makes both sides of
    # the boundary share ONE HIP runtime (two runtimes in a process cannot see each other's device pointers)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    fp, vp, i = C.c_void_p, C.c_void_p, C.c_int
    lib.cf_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(cf_config)]
    lib.cf_create.restype = i
    lib.cf_destroy.argtypes = [vp]
    lib.cf_destroy.restype = None
    lib.cf_last_error.argtypes = [vp]
    lib.cf_last_error.restype = C.c_char_p
    lib.cf_workspace_bytes.argtypes = [vp]
    lib.cf_workspace_bytes.restype = C.c_size_t
    lib.cf_load_weights.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), i]
    lib.cf_load_weights.restype = i
    lib.cf_finalize_weights.argtypes = [vp, vp]
    lib.cf_finalize_weights.restype = i
    lib.cf_warp.argtypes = [vp, fp, fp, fp, i, i, i, i, i, i, i, vp]
    lib.cf_warp.restype = i
    lib.cf_cista_forward.argtypes = [vp] + [fp] * 11 + [vp]
    lib.cf_cista_forward.restype = i
    lib.cf_flow_forward.argtypes = [vp] + [fp] * 6 + [vp]
    lib.cf_flow_forward.restype = i
    lib.cf_step.argtypes = [vp] + [fp] * 18 + [vp]
    lib.cf_step.restype = i
    lib.cf_op_conv2d.argtypes = [fp, i, i, i, i, fp, fp, i, i, i, i, i, i, i, i, i, i, fp, vp]
    lib.cf_op_conv2d.restype = i
    lib.cf_op_conv2d_bench.argtypes = [fp, i, i, i, i, fp, fp, i, i, i, i, i, i, i, i, i, i, fp, vp, i, C.POINTER(C.c_float), i]
    lib.cf_op_conv2d_bench.restype = i
    if hasattr(lib, "cf_op_conv2d_inorm_stats"):    # absent only from older tuning builds loaded through CF_LIB_PATH
        lib.cf_op_conv2d_inorm_stats.argtypes = [fp, i, i, i, i, fp, fp, i, i, i, i, i, i, i, i, i, fp, fp, C.c_float, vp]
        lib.cf_op_conv2d_inorm_stats.restype = i
    lib.cf_op_instance_norm_relu.argtypes = [fp, fp, i, i, i, i, C.c_float, vp]
    lib.cf_op_instance_norm_relu.restype = i
    lib.cf_op_corr_lookup.argtypes = [fp, fp, fp, fp, i, i, i, i, vp]
    lib.cf_op_corr_lookup.restype = i
    lib.cf_op_nchw_to_nhwc.argtypes = [fp, fp, i, i, i, i, vp]
    lib.cf_op_nchw_to_nhwc.restype = i
    lib.cf_op_nhwc_to_nchw.argtypes = [fp, fp, i, i, i, i, vp]
    lib.cf_op_nhwc_to_nchw.restype = i
    lib.cf_events_to_voxel.argtypes = [fp, fp, i, i, i, i, fp, fp, i, vp]
    lib.cf_events_to_voxel.restype = i
    lib.cf_events_to_voxel_ex.argtypes = [fp, fp, i, i, i, i, fp, fp, i, C.c_float, vp]
    lib.cf_events_to_voxel_ex.restype = i
    lib.cf_voxel_preprocess.argtypes = [fp, i, C.c_longlong, fp, i, C.c_float, vp]
    lib.cf_voxel_preprocess.restype = i
    lib.cf_quantize_u8.argtypes = [fp, fp, C.c_longlong, vp]
    lib.cf_quantize_u8.restype = i
    lib.cf_metrics_scratch_doubles.argtypes = []
    lib.cf_metrics_scratch_doubles.restype = C.c_size_t
    lib.cf_metrics_recon.argtypes = [fp, fp, C.c_longlong, fp, fp, vp]
    lib.cf_metrics_recon.restype = i
    lib.cf_metrics_flow.argtypes = [fp, fp, fp, fp, fp, i, i, i, i, C.c_float, fp, fp, vp]
    lib.cf_metrics_flow.restype = i
    lib.cf_metrics_fwl.argtypes = [fp, fp, i, i, i, i, fp, fp, vp]
    lib.cf_metrics_fwl.restype = i
    lib.cf_metrics_ssim.argtypes = [fp, fp, i, i, i, fp, fp, vp]
    lib.cf_metrics_ssim.restype = i
    lib.cf_graph_enable.argtypes = [vp, i]
    lib.cf_graph_enable.restype = i
    lib.cf_graph_stats.argtypes = [vp, C.POINTER(C.c_longlong)]
    lib.cf_graph_stats.restype = i
    lib.cf_hint_prev_grid.argtypes = [vp, i]
    lib.cf_hint_prev_grid.restype = i
    lib.cf_profile_enable.argtypes = [vp, i]
    lib.cf_profile_enable.restype = i
    lib.cf_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong), i]
    lib.cf_profile_read.restype = i
    lib.cf_profile_report.argtypes = [vp]
    lib.cf_profile_report.restype = C.c_char_p
    lib.cf_profile_report_json.argtypes = [vp]
    lib.cf_profile_report_json.restype = C.c_char_p
    lib.cf_conv_tile_name.argtypes = [i]
    lib.cf_conv_tile_name.restype = C.c_char_p
    lib.cf_conv_tile_mfma_ratio.argtypes = [i]
    lib.cf_conv_tile_mfma_ratio.restype = C.c_double
    lib.cf_plan_enable.argtypes = [vp, i]
    lib.cf_plan_enable.restype = i
    lib.cf_plan_json.argtypes = [vp]
    lib.cf_plan_json.restype = C.c_char_p
    lib.cf_conv_plan.argtypes = [C.POINTER(C.c_int), i, C.POINTER(C.c_int)]
    lib.cf_conv_plan.restype = i
    lib.cf_flow_to_bgr.argtypes = [fp, i, i, i, vp, vp, vp]
    lib.cf_flow_to_bgr.restype = i
    _lib = lib
    return lib


def ptr(t):
    """Raw device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def conv_plan(desc):
    """The launcher's tile choice for one recorded descriptor (Handle.plan() row["desc"]): pure host logic, no GPU. -> (tile, kernel)"""
    lib = load()
    arr = (C.c_int * len(desc))(*[int(v) for v in desc])
    tile = C.c_int(0)
    rc = lib.cf_conv_plan(arr, len(desc), C.byref(tile))
    if rc != 0:
        raise RuntimeError("cf_conv_plan rejected the descriptor (%d)" % rc)
    return tile.value, lib.cf_conv_tile_name(tile.value).decode()


def current_stream_ptr(device=None):
    """torch's current stream ON `device` (a tensor's device, not the process's current device: with the model on
    cuda:1 and current device cuda:0 the two differ as soon as a non-default stream is in use)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def check_f32_cuda(t, name, shape=None):
    import torch
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (cista_flow_amd has no CPU path)" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s: expected shape %s, got %s" % (name, tuple(shape), tuple(t.shape)))


class Handle:
    """Owns one cf_handle (workspace arena + packed weights) for a fixed (mode, B, H, W)."""

    def __init__(self, mode, batch, height, width, num_bins=5, base_channels=64, depth=5, iters=6,
                 warp_mode=CF_WARP_FORWARD, device=0, precision=0):
        self.lib = load()
        self.cfg = cf_config(mode, batch, height, width, num_bins, base_channels, depth, iters, warp_mode, device,
                             precision)
        h = C.c_void_p()
        rc = self.lib.cf_create(C.byref(h), C.byref(self.cfg))
        if rc != 0:
            raise RuntimeError("cf_create failed (%d): %s" % (rc, self.lib.cf_last_error(None).decode()))
        self.h = h
        self._keep = []

    def check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.cf_last_error(self.h).decode()))

    def load_state_dict(self, sd):
        """Announce every fp32 tensor of a reference-layout state_dict and pack on the current stream."""
        import torch
        keep = []
        for k, v in sd.items():
            if not isinstance(v, torch.Tensor) or v.dtype != torch.float32:
                continue   # num_batches_tracked (int64) is not used in eval
            t = v.detach().contiguous()
            check_f32_cuda(t, k)
            keep.append(t)
            shp = (C.c_int64 * max(1, t.dim()))(*t.shape)
            self.check(self.lib.cf_load_weights(self.h, k.encode(), ptr(t), shp, t.dim()), "cf_load_weights(%s)" % k)
        self.check(self.lib.cf_finalize_weights(self.h, current_stream_ptr(self.cfg.device)), "cf_finalize_weights")
        del keep

    def graph_enable(self, on=True):
        """hipGraph replay of cf_step (opt-in; CF_GRAPH=1 sets the default)."""
        self.check(self.lib.cf_graph_enable(self.h, 1 if on else 0), "cf_graph_enable")

    def graph_stats(self):
        """-> (captures, replays, executables cached)."""
        out = (C.c_longlong * 3)()
        self.check(self.lib.cf_graph_stats(self.h, out), "cf_graph_stats")
        return int(out[0]), int(out[1]), int(out[2])

    def profile_enable(self, on=True):
        self.check(self.lib.cf_profile_enable(self.h, 1 if on else 0), "cf_profile_enable")

    def profile_read(self):
        """-> list of dicts per conv tile kind: name, ms, flops, count (index 0 = all conv launches)."""
        n = 56
        ms, fl, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_longlong * n)()
        self.check(self.lib.cf_profile_read(self.h, ms, fl, cnt, n), "cf_profile_read")
        out = []
        for t in range(n):
            name = "conv_*_kernel<*>" if t == 0 else self.lib.cf_conv_tile_name(t).decode()
            out.append(dict(name=name, ms=ms[t], flops=fl[t], count=int(cnt[t])))
        return out

    def profile_report(self):
        return self.lib.cf_profile_report(self.h).decode()

    def profile_rows(self):
        """Rows of the last profile_read over both roofline classes: tag, kernel, grid, class, launches, ms, work."""
        import json
        return json.loads(self.lib.cf_profile_report_json(self.h).decode())

    def plan_enable(self, on=True):
        """Record, for every convolution launch from now on, the descriptor fields the tile choice depends on and the tile taken."""
        self.check(self.lib.cf_plan_enable(self.h, 1 if on else 0), "cf_plan_enable")

    def plan(self):
        """-> {"fields": [...], "rows": [{"tag", "tile", "kernel", "desc"}, ...]} (deduplicated) since plan_enable()."""
        import json
        return json.loads(self.lib.cf_plan_json(self.h).decode())

    @property
    def workspace_bytes(self):
        return int(self.lib.cf_workspace_bytes(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.cf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
