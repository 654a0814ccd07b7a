"""ORACLE -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

CPU restatement of the reference's Ultra.forward as plain functions over a state_dict, using the
C oracle (oracle/rspmm_oracle.c) for the relational SpMM and the reference's own node-major data
flow (transpose -> rspmm on (N, batch * dim) -> transpose back).  Follows, line by line:

  Ultra.forward                          /root/reference/ultra/models.py:18-26
  RelNBFNet.bellmanford / forward        /root/reference/ultra/models.py:55-102
  EntityNBFNet.bellmanford / forward     /root/reference/ultra/models.py:131-209
  BaseNBFNet.negative_sample_to_tail     /root/reference/ultra/base_nbfnet.py:79-86
  GeneralizedRelationalConv.forward / message_and_aggregate / update
                                         /root/reference/ultra/layers.py:67-88, 183-231, 233-240

Parity is PINNED: tests/test_oracle_model.py compares it with the committed golden scores that
tests/golden/gen_golden.py recorded from the unchanged reference modules + shipped checkpoints, and
(where /root/reference exists) with the live reference.  Only the fused sum / mean / max paths of
distmult / transe are restated (the BASELINE configs); eval mode only.
"""
import torch
from torch.nn import functional as F

from . import rspmm_oracle

MESSAGE2MUL = {"transe": "add", "distmult": "mul"}


def _rspmm(rspmm_fn, edge_index, edge_type, edge_weight, relation, input, sum, mul):
    if rspmm_fn is not None:
        return rspmm_fn(edge_index, edge_type, edge_weight, relation, input, sum=sum, mul=mul)
    return rspmm_oracle.generalized_rspmm(edge_index, edge_type, edge_weight, relation, input, sum=sum, mul=mul)


def conv_layer(sd, prefix, input, relation, boundary, edge_index, edge_type, num_node, message_func, aggregate_func,
               layer_norm=True, rspmm_fn=None):
    """One GeneralizedRelationalConv.forward on (batch, N, d) tensors; `relation` is (batch, R, d)."""
    batch_size = input.shape[0]
    edge_weight = torch.ones(edge_index.shape[1], dtype=input.dtype)  # layers.py:81-82
    # message_and_aggregate, layers.py:189-230
    x = input.transpose(0, 1).flatten(1)
    rel = relation.transpose(0, 1).flatten(1)
    bnd = boundary.transpose(0, 1).flatten(1)
    mul = MESSAGE2MUL[message_func]
    if aggregate_func == "sum":
        update = _rspmm(rspmm_fn, edge_index, edge_type, edge_weight, rel, x, "add", mul) + bnd
    elif aggregate_func == "mean":
        degree_out = torch.bincount(edge_index[1], minlength=num_node).to(input.dtype).unsqueeze(-1) + 1   # layers.py:193
        update = (_rspmm(rspmm_fn, edge_index, edge_type, edge_weight, rel, x, "add", mul) + bnd) / degree_out
    elif aggregate_func == "max":
        update = torch.max(_rspmm(rspmm_fn, edge_index, edge_type, edge_weight, rel, x, "max", mul), bnd)
    else:
        raise ValueError("oracle restates sum / mean / max only, got `%s`" % aggregate_func)
    update = update.view(num_node, batch_size, -1).transpose(0, 1)
    # update(), layers.py:233-240
    output = F.linear(torch.cat([input, update], dim=-1), sd[prefix + "linear.weight"], sd[prefix + "linear.bias"])
    if layer_norm:
        output = F.layer_norm(output, (output.shape[-1],), sd[prefix + "layer_norm.weight"],
                              sd[prefix + "layer_norm.bias"])
    return F.relu(output)


def _num_layers(sd, prefix):
    n = 0
    while (prefix + "layers.%d.linear.weight" % n) in sd:
        n += 1
    return n


def rel_nbfnet(sd, rel_graph, query_rels, cfg, rspmm_fn=None):
    """RelNBFNet.forward: (batch,) query relation ids -> (batch, num_rel, dim). models.py:55-102."""
    prefix = "relation_model."
    dim = sd[prefix + "layers.0.relation.weight"].shape[1]
    batch_size = len(query_rels)
    num_node = rel_graph.num_nodes
    query = torch.ones(batch_size, dim, dtype=sd[prefix + "layers.0.relation.weight"].dtype)
    index = query_rels.unsqueeze(-1).expand_as(query)
    boundary = torch.zeros(batch_size, num_node, dim, dtype=query.dtype)
    boundary.scatter_add_(1, index.unsqueeze(1), query.unsqueeze(1))
    layer_input = boundary
    for i in range(_num_layers(sd, prefix)):
        lp = prefix + "layers.%d." % i
        relation = sd[lp + "relation.weight"].expand(batch_size, -1, -1)                       # layers.py:76
        hidden = conv_layer(sd, lp, layer_input, relation, boundary, rel_graph.edge_index, rel_graph.edge_type,
                            num_node, cfg["message_func"], cfg["aggregate_func"], cfg.get("layer_norm", True), rspmm_fn)
        if cfg.get("short_cut", False) and hidden.shape == layer_input.shape:
            hidden = hidden + layer_input
        layer_input = hidden
    return layer_input


def negative_sample_to_tail(h_index, t_index, r_index, num_direct_rel):
    is_t_neg = (h_index == h_index[:, [0]]).all(dim=-1, keepdim=True)
    new_h = torch.where(is_t_neg, h_index, t_index)
    new_t = torch.where(is_t_neg, t_index, h_index)
    new_r = torch.where(is_t_neg, r_index, r_index + num_direct_rel)
    return new_h, new_t, new_r


def entity_nbfnet(sd, data, relation_representations, batch, cfg, rspmm_fn=None):
    """EntityNBFNet.forward in eval mode. models.py:131-209."""
    prefix = "entity_model."
    h_index, t_index, r_index = batch.unbind(-1)
    shape = h_index.shape
    h_index, t_index, r_index = negative_sample_to_tail(h_index, t_index, r_index, data.num_relations // 2)
    assert (h_index[:, [0]] == h_index).all() and (r_index[:, [0]] == r_index).all()
    h0, r0 = h_index[:, 0], r_index[:, 0]
    batch_size = len(r0)
    num_node = data.num_nodes
    query = relation_representations[torch.arange(batch_size), r0]
    index = h0.unsqueeze(-1).expand_as(query)
    boundary = torch.zeros(batch_size, num_node, query.shape[-1], dtype=query.dtype)
    boundary.scatter_add_(1, index.unsqueeze(1), query.unsqueeze(1))
    layer_input = boundary
    for i in range(_num_layers(sd, prefix)):
        lp = prefix + "layers.%d." % i
        # relation_projection: Linear - ReLU - Linear on the relation representations (layers.py:80)
        relation = F.linear(F.relu(F.linear(relation_representations, sd[lp + "relation_projection.0.weight"],
                                            sd[lp + "relation_projection.0.bias"])),
                            sd[lp + "relation_projection.2.weight"], sd[lp + "relation_projection.2.bias"])
        hidden = conv_layer(sd, lp, layer_input, relation, boundary, data.edge_index, data.edge_type, num_node,
                            cfg["message_func"], cfg["aggregate_func"], cfg.get("layer_norm", True), rspmm_fn)
        if cfg.get("short_cut", False) and hidden.shape == layer_input.shape:
            hidden = hidden + layer_input
        layer_input = hidden
    node_query = query.unsqueeze(1).expand(-1, num_node, -1)
    feature = torch.cat([layer_input, node_query], dim=-1)
    index = t_index.unsqueeze(-1).expand(-1, -1, feature.shape[-1])
    feature = feature.gather(1, index)
    hidden = F.relu(F.linear(feature, sd[prefix + "mlp.0.weight"], sd[prefix + "mlp.0.bias"]))
    score = F.linear(hidden, sd[prefix + "mlp.2.weight"], sd[prefix + "mlp.2.bias"]).squeeze(-1)
    return score.view(shape)


def ultra_forward(state_dict, cfg_like, data, batch, rspmm_fn=None):
    """Ultra.forward (models.py:18-26).  cfg_like = {"rel_model_cfg": {...}, "entity_model_cfg": {...}}
    (only message_func / aggregate_func / short_cut / layer_norm are read).  rspmm_fn lets the caller
    swap in the reference's own compiled kernel (oracle/_ref) for the C restatement."""
    sd = {k: v.detach().cpu() for k, v in state_dict.items()}
    data = data.to("cpu") if hasattr(data, "to") else data
    batch = batch.cpu()
    with torch.no_grad():
        query_rels = batch[:, 0, 2]
        rel_repr = rel_nbfnet(sd, data.relation_graph, query_rels, cfg_like["rel_model_cfg"], rspmm_fn)
        return entity_nbfnet(sd, data, rel_repr, batch, cfg_like["entity_model_cfg"], rspmm_fn)


def reference_rspmm_fn():
    """generalized_rspmm backed by the reference's own compiled CPU kernel (oracle/_ref), or None."""
    from . import build_ref
    if not build_ref.available():
        return None
    mod = build_ref.load()

    def fn(edge_index, edge_type, edge_weight, relation, input, sum="add", mul="mul"):
        ei, et, ew, _ = rspmm_oracle.sort_edges(edge_index, edge_type, edge_weight)
        return getattr(mod, "rspmm_%s_%s_forward_cpu" % (sum, mul))(ei, et, ew, relation.contiguous(),
                                                                    input.contiguous())
    return fn
