"""mspi_amd.runtime.GraphPipeline: hipGraph capture + batches in flight, on a small stand-in forward (the full model goes
through it in tests/test_inference.py and bench.py)."""
import pytest
import torch


def test_pipeline_needs_device_inputs():
    from mspi_amd.runtime import GraphPipeline
    with pytest.raises(ValueError, match="device-resident"):
        GraphPipeline(lambda x: x, (torch.zeros(4),))


@pytest.mark.gpu
def test_pipeline_replays_match_eager(dev):
    from mspi_amd import engine as E
    from mspi_amd.runtime import GraphPipeline
    g = torch.Generator().manual_seed(0)
    w = torch.randn(64, 32, 1, 1, 1, generator=g)
    pk = E.pack_conv(w, torch.randn(64, generator=g), act=E.ACT_RELU, device=dev)

    def fn(x):                                   # [N,32,1,H,W] -> two outputs, through the C ABI
        y = E.conv(x, pk)
        return y.as_rows().sum(1), E.postprocess_u8(y.as_rows()[:, :1].reshape(2, 24, 40).contiguous(), (48, 80))

    xs = [torch.randn(2, 32, 1, 24, 40, generator=g).to(dev) for _ in range(5)]
    want = [tuple(t.clone() for t in fn(x)) for x in xs]
    pipe = GraphPipeline(fn, (xs[0],), depth=2, layouts=2)
    assert pipe.depth == 2 and pipe.layout in (0, 1)
    tickets = []
    for i, x in enumerate(xs):                   # two in flight: fetch batch i-1 after submitting batch i
        tickets.append(pipe.submit(x))
        if i:
            got = pipe.fetch(tickets[i - 1])
            assert torch.equal(got[0], want[i - 1][0]) and torch.equal(got[1], want[i - 1][1])
    host = pipe.fetch_host(tickets[-1])
    assert not host[1].is_cuda and host[1].is_pinned() and torch.equal(host[1], want[-1][1].cpu())
    assert torch.equal(host[0], want[-1][0].cpu())
    # resident-input form replays the last submitted inputs of that slot
    t = pipe.submit()
    assert torch.equal(pipe.fetch(t)[0], want[3][0])          # slot 1 last held xs[3]
    # input assembly on the slot's own stream
    t = pipe.submit_build(lambda ins: ins[0].copy_(xs[2]))
    assert torch.equal(pipe.fetch(t)[0], want[2][0])
    with pytest.raises(ValueError, match="captured for"):
        pipe.submit(xs[0][:1])
    idle = pipe.idle_streams(2, candidates=6)
    assert len(idle) == 2 and len(pipe.idle_latency_ms) == 6 and pipe.idle_latency_ms[0] <= pipe.idle_latency_ms[-1]
    assert pipe.latency_ms(3) > 0
    pipe.drain()
