"""Handle cache shared by the nn.Module shells: one cf_handle per (batch, device), weights re-packed
whenever the module's parameters change (load_state_dict, .to(), in-place edits)."""
import torch

from . import lib as _lib

# Registration epoch: bumped whenever ANY nn.Module registers a parameter, buffer or sub-module (assigning a new
# nn.Parameter / module goes through these hooks).  A backend re-collects its tensor list only when the epoch moved;
# per frame it then compares (data_ptr, _version) of that cached list (~40 us for the 262 tensors of cista-eiflow
# instead of ~360 us for a state_dict() walk): .to() / .cuda() change data_ptr, load_state_dict and in-place edits
# bump _version.  A plain tensor assigned over an existing BUFFER attribute (module.running_mean = t) or a direct edit of
# module._parameters / _buffers goes through no hook: those are caught by the full re-collection every REWALK_EVERY frames
# (a stale window of at most that many frames; amortised cost ~6 us per frame) -- or at once with backend.invalidate().
_EPOCH = [0]
REWALK_EVERY = 64


def _bump(*_args, **_kw):
    _EPOCH[0] += 1
    return None


try:
    from torch.nn.modules import module as _nnmod
    _nnmod.register_module_parameter_registration_hook(_bump)
    _nnmod.register_module_buffer_registration_hook(_bump)
    _nnmod.register_module_module_registration_hook(_bump)
    _HOOKS = True
except Exception:      # very old torch: fall back to the full walk every frame
    _HOOKS = False


class HipBackend(object):
    def __init__(self, module, mode, image_dim, num_bins=5, base_channels=64, depth=5, iters=6, warp_mode='forward'):
        import os
        # arithmetic of the conv products: module.precision ("f32" default | "f16x3" | "f16"), env CF_PRECISION overrides
        self.precision = os.environ.get("CF_PRECISION") or getattr(module, "precision", "f32")
        if self.precision not in _lib.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(_lib.PRECISIONS))
        self.module = module
        self.mode = mode
        self.image_dim = (int(image_dim[0]), int(image_dim[1]))
        self.kw = dict(num_bins=num_bins, base_channels=base_channels, depth=depth, iters=iters,
                       warp_mode=_lib.CF_WARP_FORWARD if warp_mode == 'forward' else _lib.CF_WARP_BACKWARD)
        self.handles = {}     # (B, device index) -> [Handle, weight signature]
        self._tensors = None  # cached state_dict tensors (see _EPOCH above)
        self._epoch = -1
        self._age = 0         # frames since the tensor list was collected

    def _signature(self):
        self._age += 1
        if not _HOOKS or self._tensors is None or self._epoch != _EPOCH[0] or self._age >= REWALK_EVERY:
            # a fresh list drops the references to tensors the module no longer holds
            self._tensors = list(self.module.state_dict(keep_vars=True).values())
            self._epoch = _EPOCH[0]
            self._age = 0
        return tuple([(t.data_ptr(), t._version) for t in self._tensors])

    def invalidate(self):
        """Force a re-pack of the weights on the next forward (after editing module._buffers / _parameters by hand or
        assigning a plain tensor over a buffer attribute; everything else is detected automatically)."""
        self._tensors = None
        for ent in self.handles.values():
            ent[1] = None

    def get(self, batch, device):
        if device.type != 'cuda':
            raise RuntimeError("cista_flow_amd runs on the GPU only (got device %s); move the model and inputs with "
                               ".to('cuda')" % device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = (int(batch), idx)
        ent = self.handles.get(key)
        if ent is None:
            h = _lib.Handle(self.mode, int(batch), self.image_dim[0], self.image_dim[1], device=idx,
                            precision=_lib.PRECISIONS[self.precision], **self.kw)
            ent = [h, None]
            self.handles[key] = ent
        sig = self._signature()
        if ent[1] != sig:
            sd = self.module.state_dict()
            for k, v in sd.items():
                if isinstance(v, torch.Tensor) and v.is_floating_point() and (not v.is_cuda or v.device.index != idx):
                    raise RuntimeError("parameter %s lives on %s but the input is on cuda:%d" % (k, v.device, idx))
            with torch.cuda.device(idx):
                ent[0].load_state_dict(sd)
            ent[1] = sig
        return ent[0]

    def clear(self):
        for h, _ in self.handles.values():
            h.close()
        self.handles = {}


def nhwc_state(t, name, shape):
    """A recurrent state as NHWC memory: channels_last tensors pass through untouched."""
    _lib.check_f32_cuda(t, name, shape)
    return t.contiguous(memory_format=torch.channels_last)


def empty_nhwc(B, C, H, W, device):
    return torch.empty((B, C, H, W), dtype=torch.float32, device=device, memory_format=torch.channels_last)
