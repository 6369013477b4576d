"""Out-of-bounds canaries for the GPU tests ("out-of-bounds canaries around every output buffer in tests").

`GuardedAllocations` replaces torch.empty / empty_like / zeros / zeros_like for CUDA tensors while it is active: every device
buffer the product allocates becomes a view into a larger one with a 512-byte halo on both sides.  Halo AND body are pre-filled
with 0xFF bytes -- a NaN in f32 / bf16, -1 in integers -- so that
  * a kernel that WRITES outside its output (row maps, halo tiles, ragged last tiles are where an off-by-one hides) breaks a
    halo, which verify() reports with the allocation's shape and call site;
  * a kernel that READS outside its input, or leaves part of its output unwritten, feeds NaNs into results the parity
    assertions then reject.
The conftest fixture turns it on for every `-m gpu` test (opt out: @pytest.mark.noguard, used by timing tests).
"""
import traceback

import torch

HALO = 512          # bytes, a multiple of the 256-byte alignment the kernels may assume
BODY_FILL_LIMIT = 256 << 20


class GuardedAllocations:
    def __init__(self):
        self.records = []
        self._orig = {}

    # -- allocation ------------------------------------------------------------------------------------------------------
    @staticmethod
    def _is_cuda(device):
        if device is None:
            return False
        return torch.device(device).type == "cuda"

    def _alloc(self, size, dtype, device, zero=False):
        dtype = dtype or torch.get_default_dtype()
        n = 1
        for s in size:
            n *= int(s)
        es = torch.empty(0, dtype=dtype).element_size()
        halo_e = HALO // es
        raw = self._orig["empty"](n + 2 * halo_e, dtype=dtype, device=device)
        u8 = raw.view(torch.uint8)
        u8[:HALO].fill_(0xFF); u8[HALO + n * es:].fill_(0xFF)
        body = raw[halo_e:halo_e + n]
        if zero:
            body.zero_()
        elif n * es <= BODY_FILL_LIMIT:
            u8[HALO:HALO + n * es].fill_(0xFF)
        where = "".join(traceback.format_stack(limit=6)[:-2][-3:])
        self.records.append((u8, n * es, tuple(size), dtype, where))
        return body.view(*size) if len(size) else body.view(())

    @staticmethod
    def _size(args):
        if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)):
            return tuple(args[0])
        return tuple(args)

    def _wrap_new(self, name, zero):
        orig = self._orig[name]

        def fn(*args, dtype=None, device=None, **kw):
            if self._is_cuda(device) and not kw:      # anything unusual (requires_grad, out=, pin_memory, ...) passes through
                return self._alloc(self._size(args), dtype, device, zero)
            return orig(*args, dtype=dtype, device=device, **kw)
        return fn

    def _wrap_like(self, name, zero):
        orig = self._orig[name]

        def fn(t, *args, dtype=None, device=None, **kw):
            dev = device if device is not None else t.device
            if self._is_cuda(dev) and t.is_contiguous() and not args and not kw:
                return self._alloc(tuple(t.shape), dtype or t.dtype, dev, zero)
            return orig(t, *args, dtype=dtype, device=device, **kw)
        return fn

    def __enter__(self):
        for name in ("empty", "zeros", "empty_like", "zeros_like"):
            self._orig[name] = getattr(torch, name)
        torch.empty = self._wrap_new("empty", False)
        torch.zeros = self._wrap_new("zeros", True)
        torch.empty_like = self._wrap_like("empty_like", False)
        torch.zeros_like = self._wrap_like("zeros_like", True)
        return self

    def __exit__(self, *exc):
        for name, fn in self._orig.items():
            setattr(torch, name, fn)
        return False

    # -- verification ------------------------------------------------------------------------------------------------------
    def verify(self):
        """Synchronise and check every halo; raises AssertionError naming the first broken allocation."""
        torch.cuda.synchronize()
        broken = []
        for u8, nbytes, size, dtype, where in self.records:
            lo_ok = bool((u8[:HALO] == 0xFF).all())
            hi_ok = bool((u8[HALO + nbytes:] == 0xFF).all())
            if not (lo_ok and hi_ok):
                side = ("below" if not lo_ok else "") + ("/" if not lo_ok and not hi_ok else "") + ("above" if not hi_ok else "")
                broken.append(f"out-of-bounds write {side} a {dtype} buffer of shape {size}, allocated at:\n{where}")
        n = len(self.records)
        self.records.clear()
        assert not broken, f"{len(broken)} of {n} guarded buffers damaged:\n" + "\n".join(broken[:3])
        return n
