"""Do CU-masked streams (hipExtStreamCreateWithCUMask) keep the entity layers of one batch and the short launches of the other apart?

Two batches in flight share the chip by launch size today (192 workgroups of an entity layer, the other batch's launches on what is
left) -- but a launch that finds CUs free takes them: a relation-graph layer or a readout of batch B that starts in a gap of batch
A's entity layers spreads over all 256 CUs, and A's next entity layer (one workgroup per CU, 160 KB of LDS) waits for it: as timed the
entity layer takes 131 us on average against 99 alone.  Masks would pin the two kinds of work to disjoint CUs.  Measured here:
  A  the entity layer (192 workgroups) on a stream masked to 192 CUs against an unmasked stream;
  B  entity layers on the 192-CU stream WHILE relation-graph layers run on a 64-CU stream, against both on unmasked streams;
  C  does a captured graph replayed on a masked stream keep the mask (20 relation-graph layers: 240 workgroups on 64 CUs)?
"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ultra_amd import layers, rspmm, synthetic  # noqa: E402

dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]


def masked_stream(lo, hi):
    words = (ctypes.c_uint32 * 8)()
    for b in range(lo, hi):
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, "hipExtStreamCreateWithCUMask -> %d" % rc
    return torch.cuda.ExternalStream(s.value, device=dev)


data = synthetic.make_kg(**synthetic.SHAPES["fb15k237"], seed=1234)
N, R = data.num_nodes, int(data.num_relations)
g = torch.Generator().manual_seed(0)
bs = 8
x = torch.randn(bs, N, 64, generator=g).to(dev)
rel = torch.randn(bs, R, 64, generator=g).to(dev)
point = (torch.arange(bs, device=dev) * 7 % N, torch.randn(bs, 64, generator=g).to(dev))
w = (torch.randn(64, 128, generator=g) / 11).to(dev)
b, lw, lb = (torch.randn(64, generator=g).to(dev) for _ in range(3))
plan = rspmm.Plan(data.edge_index, data.edge_type, N, R, exact_order=True)
rg = data.relation_graph
rplan = rspmm.Plan(rg.edge_index.to(dev) if False else rg.edge_index, rg.edge_type, rg.num_nodes, 4, exact_order=True)
rlayer = layers.GeneralizedRelationalConv(64, 64, 4, 64, "distmult", "sum", True, "relu").to(dev)
rx = torch.randn(bs, rg.num_nodes, 64, device=dev)
rrel = torch.randn(bs, 4, 64, device=dev)


def entity():
    rspmm.set_tuning(update_form=3, grid=192)
    out = plan.forward_update(rel, x, w, b, lw, lb, 1e-5, 7, point=point)
    rspmm.set_tuning()
    return out


def relation():
    return rplan.fused_layer(rrel, rx, rlayer.linear, rlayer.layer_norm, relu=True, residual=True)


def graph_of(fn, n, stream):
    fn()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    return gr


def time_graph(gr, stream, reps=5):
    with torch.cuda.stream(stream):
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            gr.replay()
        e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


plain_a, plain_b = torch.cuda.Stream(), torch.cuda.Stream()
m192, m64 = masked_stream(0, 192), masked_stream(192, 256)
m_all = masked_stream(0, 256)
ge = graph_of(entity, 10, plain_a)
gr = graph_of(relation, 20, plain_b)
print("A  ten entity layers (192 workgroups): unmasked stream %.1f us per layer | 192-CU mask %.1f | 256-CU mask %.1f | 64-CU mask %.1f" %
      (time_graph(ge, plain_a) / 10, time_graph(ge, m192) / 10, time_graph(ge, m_all) / 10, time_graph(ge, m64) / 10))
print("C  twenty relation-graph layers (240 workgroups): unmasked %.1f us per layer | 64-CU mask %.1f | 192-CU mask %.1f" %
      (time_graph(gr, plain_b) / 20, time_graph(gr, m64) / 20, time_graph(gr, m192) / 20))


def together(se, sr, rounds=6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(se):
        e0.record(se)
    for _ in range(rounds):
        with torch.cuda.stream(sr):
            gr.replay()
            gr.replay()
        with torch.cuda.stream(se):
            ge.replay()
    with torch.cuda.stream(se):
        e1.record(se)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / rounds * 1e3
    return e0.elapsed_time(e1) / rounds / 10 * 1e3, wall


for name, se, sr in (("unmasked streams", plain_a, plain_b), ("entity on the 192-CU mask, relation layers on the 64-CU mask", m192, m64)):
    us, wall = together(se, sr)
    us, wall = together(se, sr)
    print("B  %s: entity layer %.1f us while relation-graph layers run beside it; one round (10 entity + 40 relation layers) %.3f ms" % (name, us, wall))
