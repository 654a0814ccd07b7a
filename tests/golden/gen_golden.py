"""Generate the committed golden vectors by RUNNING THE REFERENCE in this container.

    python tests/golden/gen_golden.py

Needs /root/reference (so it only runs in the build container, never on the GPU box).  The unchanged
reference modules (ultra.rspmm, ultra.layers, ultra.models, ultra.tasks) are imported under the
test-only torch_geometric / torch_scatter shim in tests/golden/pyg_shim/ and executed on CPU with
the shipped checkpoints.  Outputs (small, committed):

  rspmm_<sum>_<mul>.pt      inputs, output and gradients of the reference generalized_rspmm (rspmm.py:168)
  ultra_3g_model.pt         ckpts/ultra_3g.pth["model"]  (the state dict only, 168,705 fp32 values)
  ultra_50g_model.pt        ckpts/ultra_50g.pth["model"]
  model_<ckpt>_<aggr>.pt    a seeded small KG, its reference relation graph, all-negative batches, reference
                            scores for tail and head batches, filter masks and rankings
  query_nbfnet_ultra_3g.pt  QueryNBFNet (models.py:212-275) on dense initial node features
  relation_projection_ultra_3g.pt   RelationProjection (ultraquery.py:245-277) with and without its threshold
  negative_sampling.pt      tasks.negative_sampling (tasks.py:42-76) on a seeded KG with a hub node: for several (batch,
                            num_negative, strict) cases the global generator's state right before the call and the batch the
                            reference returned -- the tests replay the state through ultra_amd.tasks.negative_sampling (CPU: the
                            mask formulation; GPU: csrc/sampling.hip fed the same uniform draws)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
os.environ.setdefault("TORCH_EXTENSIONS_DIR", "/tmp/torch_ext_ref")
sys.path.insert(0, os.path.join(HERE, "pyg_shim"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def gen_rspmm():
    from ultra.rspmm import generalized_rspmm  # JIT-builds the reference rspmm.cpp (CPU only here)
    from tests import helpers
    for sum in ("add", "min", "max"):
        for mul in ("mul", "add"):
            ei, et = helpers.random_graph(num_node=48, num_edge=300, num_relation=5, seed=17, hub=(3, 90),
                                          empty_rows=4, duplicates=20)
            rel, x, w = helpers.features(48, 5, 40, ei.shape[1], seed=18)
            g = torch.Generator().manual_seed(19)
            og = torch.randn(48, 40, generator=g)
            w_, rel_, x_ = (t.clone().requires_grad_() for t in (w, rel, x))
            out = generalized_rspmm(ei, et, w_, rel_, x_, sum=sum, mul=mul)
            out.backward(og)
            torch.save(dict(sum=sum, mul=mul, edge_index=ei, edge_type=et, edge_weight=w, relation=rel, input=x,
                            output=out.detach(), output_grad=og, weight_grad=w_.grad, relation_grad=rel_.grad,
                            input_grad=x_.grad),
                       os.path.join(HERE, "rspmm_%s_%s.pt" % (sum, mul)))
            print("rspmm", sum, mul, "ok")


def gen_models():
    from torch_geometric.data import Data
    from ultra import tasks as ref_tasks
    from ultra.models import Ultra
    from ultra_amd import synthetic

    for ckpt_name, aggr in (("ultra_3g", "sum"), ("ultra_50g", "max")):
        ckpt = torch.load(os.path.join(REF, "ckpts", ckpt_name + ".pth"), map_location="cpu")
        state = {k: v.clone() for k, v in ckpt["model"].items()}
        torch.save(state, os.path.join(HERE, ckpt_name + "_model.pt"))

        kg = synthetic.make_kg(num_node=200, num_triple=1500, num_relation_base=6, num_test=16, seed=7,
                               relation_graph=False)
        data = Data(edge_index=kg.edge_index, edge_type=kg.edge_type, num_nodes=kg.num_nodes,
                    num_relations=kg.num_relations, target_edge_index=kg.target_edge_index,
                    target_edge_type=kg.target_edge_type)
        data = ref_tasks.build_relation_graph(data)
        cfg = synthetic.default_model_cfg(aggregate_func=aggr)
        model = Ultra(rel_model_cfg=dict(cfg["rel_model_cfg"]), entity_model_cfg=dict(cfg["entity_model_cfg"]))
        model.load_state_dict(state)
        model.eval()
        batch = kg.target_triples[:4]
        with torch.no_grad():
            t_batch, h_batch = ref_tasks.all_negative(data, batch)
            t_pred = model(data, t_batch)
            h_pred = model(data, h_batch)
            t_mask, h_mask = ref_tasks.strict_negative_mask(data, batch)
            pos_h, pos_t, pos_r = batch.t()
            t_rank = ref_tasks.compute_ranking(t_pred, pos_t, t_mask)
            h_rank = ref_tasks.compute_ranking(h_pred, pos_h, h_mask)
            # a (bs, 1 + num_neg, 3) training-style batch through the same forward
            neg_batch = ref_tasks.negative_sampling(data, batch, 8, strict=True)
            neg_pred = model(data, neg_batch)
        torch.save(dict(ckpt=ckpt_name, aggregate_func=aggr,
                        edge_index=data.edge_index, edge_type=data.edge_type, num_nodes=data.num_nodes,
                        num_relations=data.num_relations,
                        rel_edge_index=data.relation_graph.edge_index, rel_edge_type=data.relation_graph.edge_type,
                        batch=batch, t_batch=t_batch, h_batch=h_batch, t_pred=t_pred, h_pred=h_pred,
                        t_mask=t_mask, h_mask=h_mask, t_rank=t_rank, h_rank=h_rank,
                        neg_batch=neg_batch, neg_pred=neg_pred),
                   os.path.join(HERE, "model_%s_%s.pt" % (ckpt_name, aggr)))
        print("model", ckpt_name, aggr, "scores", tuple(t_pred.shape), "ranks", t_rank.tolist(), h_rank.tolist())


def gen_query_nbfnet():
    """QueryNBFNet (ultra/models.py:212-275), the UltraQuery entity reasoner: same layers as EntityNBFNet but
    initial node features and queries come from outside.  Weights: the entity_model half of ultra_3g."""
    from torch_geometric.data import Data
    from ultra import tasks as ref_tasks
    from ultra.models import QueryNBFNet, RelNBFNet
    from ultra_amd import synthetic

    state = torch.load(os.path.join(HERE, "ultra_3g_model.pt"))
    kg = synthetic.make_kg(num_node=150, num_triple=1000, num_relation_base=5, num_test=8, seed=13, relation_graph=False)
    data = Data(edge_index=kg.edge_index, edge_type=kg.edge_type, num_nodes=kg.num_nodes, num_relations=kg.num_relations)
    data = ref_tasks.build_relation_graph(data)
    cfg = synthetic.default_model_cfg()
    ent_cfg = {k: v for k, v in cfg["entity_model_cfg"].items() if k != "class"}
    rel_cfg = {k: v for k, v in cfg["rel_model_cfg"].items() if k != "class"}
    qnet, rnet = QueryNBFNet(**ent_cfg), RelNBFNet(**rel_cfg)
    qnet.load_state_dict({k[len("entity_model."):]: v for k, v in state.items() if k.startswith("entity_model.")})
    rnet.load_state_dict({k[len("relation_model."):]: v for k, v in state.items() if k.startswith("relation_model.")})
    qnet.eval(), rnet.eval()
    g = torch.Generator().manual_seed(21)
    query_rels = torch.tensor([1, 4, 7])
    node_features = torch.rand(3, kg.num_nodes, 64, generator=g) * (torch.rand(3, kg.num_nodes, 1, generator=g) < 0.1)
    with torch.no_grad():
        rel_repr = rnet(data.relation_graph, query=query_rels)
        query = rel_repr[torch.arange(3), query_rels]
        score = qnet(data, node_features, rel_repr, query)
    torch.save(dict(edge_index=data.edge_index, edge_type=data.edge_type, num_nodes=data.num_nodes,
                    num_relations=data.num_relations, rel_edge_index=data.relation_graph.edge_index,
                    rel_edge_type=data.relation_graph.edge_type, query_rels=query_rels, node_features=node_features,
                    rel_repr=rel_repr, query=query, score=score), os.path.join(HERE, "query_nbfnet_ultra_3g.pt"))
    print("query_nbfnet", tuple(score.shape))


def gen_relation_projection():
    """RelationProjection (ultra/ultraquery.py:245-277): fuzzy set of head entities x query relation -> fuzzy set of tails.
    ultra/ultraquery.py pulls in the whole query stack (datasets, torch_scatter composites) at import, so only the
    class's own source is executed -- read from the reference file where it lies, nothing is copied into the repo --
    around the reference's Ultra(RelNBFNet, QueryNBFNet) with the ultra_3g weights."""
    from torch import nn
    from torch.nn import functional as F
    from torch_geometric.data import Data
    from ultra import tasks as ref_tasks
    from ultra.models import Ultra
    from ultra_amd import synthetic

    src = open(os.path.join(REF, "ultra", "ultraquery.py")).read()
    ns = {"torch": torch, "nn": nn, "F": F}
    exec(src[src.index("class RelationProjection"):src.index("class SymbolicTraversal")], ns)
    state = torch.load(os.path.join(HERE, "ultra_3g_model.pt"))
    kg = synthetic.make_kg(num_node=150, num_triple=1000, num_relation_base=5, num_test=8, seed=13, relation_graph=False)
    data = Data(edge_index=kg.edge_index, edge_type=kg.edge_type, num_nodes=kg.num_nodes, num_relations=kg.num_relations)
    data = ref_tasks.build_relation_graph(data)
    cfg = synthetic.default_model_cfg()
    ent_cfg = dict(cfg["entity_model_cfg"])
    ent_cfg["class"] = "QueryNBFNet"
    model = Ultra(rel_model_cfg=dict(cfg["rel_model_cfg"]), entity_model_cfg=ent_cfg)
    model.load_state_dict(state)
    model.eval()
    g = torch.Generator().manual_seed(23)
    r_index = torch.tensor([2, 0, 9, 5])
    h_prob = torch.rand(4, kg.num_nodes, generator=g) * (torch.rand(4, kg.num_nodes, generator=g) < 0.15)
    out = {}
    with torch.no_grad():
        for thr in (0.0, 0.3):
            out[thr] = ns["RelationProjection"](model, threshold=thr)(data, h_prob, r_index)
    torch.save(dict(edge_index=data.edge_index, edge_type=data.edge_type, num_nodes=data.num_nodes,
                    num_relations=data.num_relations, rel_edge_index=data.relation_graph.edge_index,
                    rel_edge_type=data.relation_graph.edge_type, h_prob=h_prob, r_index=r_index,
                    t_prob=out[0.0], t_prob_thr03=out[0.3]), os.path.join(HERE, "relation_projection_ultra_3g.pt"))
    print("relation_projection", tuple(out[0.0].shape), float((out[0.0] - out[0.3]).abs().max()))


def gen_negative_sampling():
    """The reference's sampler on a KG whose node 3 is a hub (a tenth of all triples start there: hundreds of known tails per
    (hub, relation) query, most candidates of its row filtered), strict and not, odd and even batch sizes."""
    from torch_geometric.data import Data
    from ultra import tasks as ref_tasks
    from ultra_amd import synthetic
    kg = synthetic.make_kg(num_node=400, num_triple=6000, num_relation_base=5, num_test=16, seed=31, relation_graph=False)
    ei, et = kg.edge_index.clone(), kg.edge_type.clone()
    half = ei.shape[1] // 2
    hub = torch.arange(0, half, 10)
    ei[0, hub] = 3                                    # heads of the forward edges ...
    ei[1, hub + half] = 3                             # ... and tails of their inverses
    data = Data(edge_index=ei, edge_type=et, num_nodes=kg.num_nodes, num_relations=kg.num_relations)
    triples = torch.stack([ei[0, :half], ei[1, :half], et[:half]], dim=-1)
    hub_rows = triples[triples[:, 0] == 3][:3]
    inverse = torch.stack([ei[0, half:], ei[1, half:], et[half:]], dim=-1)
    hub_as_tail = inverse[inverse[:, 1] == 3][:3]    # (head, 3, inverse relation): hundreds of known heads of (tail = 3, relation)
    cases = []
    for bs, num_negative, strict, seed in ((8, 32, True, 101), (6, 7, True, 102), (5, 16, True, 103), (8, 32, False, 104),
                                           (2, 256, True, 105)):
        g = torch.Generator().manual_seed(seed)
        pick = torch.randint(0, half, (bs,), generator=g)
        batch = triples[pick].clone()
        batch[0] = hub_rows[seed % 3]                 # a hub anchor in the tail half (first half of the batch: tasks.py:50-52) ...
        batch[-1] = hub_as_tail[seed % 3]             # ... and one in the head half
        torch.manual_seed(seed)
        state = torch.get_rng_state()
        out = ref_tasks.negative_sampling(data, batch, num_negative, strict=strict)
        cases.append(dict(batch=batch, num_negative=num_negative, strict=strict, rng_state=state, out=out))
        print("negative_sampling", bs, num_negative, strict, tuple(out.shape))
    torch.save(dict(edge_index=ei, edge_type=et, num_nodes=kg.num_nodes, num_relations=kg.num_relations, cases=cases),
               os.path.join(HERE, "negative_sampling.pt"))


def gen_easy_edges():
    """BaseNBFNet.remove_easy_edges (base_nbfnet.py:54-77) of the unchanged reference: the filtered edge list for batches of
    positives that ARE graph edges plus their strict negatives (tail and head halves), with and without remove_one_hop."""
    import types
    from torch_geometric.data import Data
    from ultra import tasks as ref_tasks
    from ultra.base_nbfnet import BaseNBFNet
    from ultra_amd import synthetic
    kg = synthetic.make_kg(num_node=300, num_triple=5000, num_relation_base=6, num_test=16, seed=41, relation_graph=False)
    ei, et = kg.edge_index.clone(), kg.edge_type.clone()
    half = ei.shape[1] // 2
    ei[:, 7] = ei[:, 5]                                # a duplicate edge (both copies go) ...
    et[7] = et[5]
    ei[:, 9] = ei[:, 5]                                # ... and the same pair under another relation (goes with remove_one_hop only)
    et[9] = (et[5] + 1) % (kg.num_relations // 2)
    data = Data(edge_index=ei, edge_type=et, num_nodes=kg.num_nodes, num_relations=kg.num_relations)
    triples = torch.stack([ei[0, :half], ei[1, :half], et[:half]], dim=-1)
    cases = []
    for bs, num_negative, one_hop, seed in ((8, 32, False, 201), (8, 32, True, 202), (5, 256, False, 203), (2, 3, True, 204),
                                            (1, 1, False, 205)):
        g = torch.Generator().manual_seed(seed)
        pick = torch.randint(0, half, (bs,), generator=g)
        pick[0] = 5
        torch.manual_seed(seed)
        batch = ref_tasks.negative_sampling(data, triples[pick], num_negative, strict=True)
        h, t, r = batch.unbind(-1)
        out = BaseNBFNet.remove_easy_edges(types.SimpleNamespace(remove_one_hop=one_hop), data, h, t, r)
        # (the filter keeps the order and identical edges go or stay together: the kept list read back as a mask over the edges)
        full = torch.cat([ei, et.unsqueeze(0)]).t().tolist()
        kept = torch.cat([out.edge_index, out.edge_type.unsqueeze(0)]).t().tolist()
        keep, j = [], 0
        for row in full:
            hit = j < len(kept) and kept[j] == row
            keep.append(hit)
            j += hit
        assert j == len(kept)
        cases.append(dict(batch=batch, remove_one_hop=one_hop, keep=torch.tensor(keep)))
        print("remove_easy_edges", bs, num_negative, one_hop, ei.shape[1], "->", out.edge_index.shape[1])
    torch.save(dict(edge_index=ei, edge_type=et, num_nodes=kg.num_nodes, num_relations=kg.num_relations, cases=cases),
               os.path.join(HERE, "easy_edges.pt"))


if __name__ == "__main__":
    assert os.path.isdir(REF), "golden generation needs the reference checkout at /root/reference"
    torch.manual_seed(0)     # negative_sampling draws from the global generator
    if "--only-relation-projection" in sys.argv:
        gen_relation_projection()
        sys.exit(0)
    if "--only-negative-sampling" in sys.argv:
        gen_negative_sampling()
        sys.exit(0)
    if "--only-easy-edges" in sys.argv:
        gen_easy_edges()
        sys.exit(0)
    gen_rspmm()
    gen_models()
    gen_query_nbfnet()
    gen_relation_projection()
    gen_negative_sampling()
    gen_easy_edges()
