"""Input hand-off: DevicePrefetcher yields exactly what the loader yields (structure, values,
order, length) — on the CPU as a pass-through, on the GPU through pinned staging buffers and a copy stream."""
import pytest
import torch

from pytorch3dunet_amd.data import DevicePrefetcher


def _batches(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(2, 1, 4, 5, 6, generator=g), [torch.rand(2, 1, 4, 5, 6, generator=g), torch.rand(2, 3, generator=g)])
            for _ in range(n)]


def test_prefetcher_cpu_passthrough():
    data = _batches(4)
    pf = DevicePrefetcher(data, "cpu")
    assert len(pf) == 4
    out = list(pf)
    assert len(out) == 4
    for (x, t), (xr, tr) in zip(out, data):
        assert torch.equal(x, xr) and isinstance(t, tuple) and torch.equal(t[0], tr[0]) and torch.equal(t[1], tr[1])


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [1, 2])
def test_prefetcher_gpu_values_and_reuse(depth):
    data = _batches(7, seed=3)
    pf = DevicePrefetcher(data, "cuda", depth=depth)
    seen = []
    for x, t in pf:
        assert x.is_cuda and t[0].is_cuda and t[1].is_cuda
        y = (x * 2).sum() + t[0].sum()  # consume on the current stream while the next copy is in flight
        seen.append((x.clone(), t[0].clone(), t[1].clone(), y))
    torch.cuda.synchronize()
    assert len(seen) == 7
    for (x, t0, t1, y), (xr, tr) in zip(seen, data):
        assert torch.equal(x.cpu(), xr) and torch.equal(t0.cpu(), tr[0]) and torch.equal(t1.cpu(), tr[1])
        assert abs(y.item() - ((xr * 2).sum() + tr[0].sum()).item()) < 1e-3
    assert len(pf._pinned) <= 3 * (depth + 1)  # staging buffers are reused, not re-allocated per batch
