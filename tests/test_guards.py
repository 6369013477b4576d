"""The canary allocator itself (CPU part: pass-through; GPU part: it does catch an out-of-bounds write)."""
import pytest
import torch

from guards import GuardedAllocations, HALO


def test_cpu_allocations_pass_through():
    with GuardedAllocations() as g:
        a = torch.empty(3, 4); b = torch.zeros((2, 2), dtype=torch.int32); c = torch.empty_like(a); d = torch.zeros_like(b)
        assert a.shape == (3, 4) and b.dtype == torch.int32 and int(b.sum()) == 0 and c.shape == a.shape and int(d.sum()) == 0
        assert not g.records
    assert torch.empty.__module__ == "torch" or callable(torch.empty)


@pytest.mark.gpu
@pytest.mark.noguard
def test_guard_catches_a_write_past_the_end_and_nan_fills_the_body(dev):
    with GuardedAllocations() as g:
        x = torch.empty(10, 8, dtype=torch.bfloat16, device=dev)
        z = torch.zeros(5, dtype=torch.float32, device=dev)
        assert x.data_ptr() % 256 == 0 and x.is_contiguous() and torch.isnan(x.float()).all() and float(z.abs().sum()) == 0
        x.fill_(1.0)
        assert g.verify() == 2                                         # in-bounds writes leave the halos alone
        y = torch.empty(16, dtype=torch.float32, device=dev)
        raw = y.as_strided((17,), (1,))                                # one element past the end
        raw[16] = 3.0
        with pytest.raises(AssertionError, match="out-of-bounds write above"):
            g.verify()
        w = torch.empty(16, dtype=torch.float32, device=dev)
        w.as_strided((1,), (1,), w.storage_offset() - 1).fill_(2.0)    # one element before the start
        with pytest.raises(AssertionError, match="out-of-bounds write below"):
            g.verify()
    assert HALO % 256 == 0
