"""RelationProjection -- the second consumer of the propagation engine (reference: ultra/ultraquery.py:245-277).

One projection step of UltraQuery: a fuzzy set of head entities `h_prob` (batch, num_nodes) and one query relation per
sample become a fuzzy set of tail entities.  Same constructor and forward signature as the reference class; `model`
is an `Ultra` whose entity model is a `QueryNBFNet` (models.py:212-275).  The query executor around it (fuzzy logic,
postfix stack, symbolic traversal: ultraquery.py:12-243, 281-298) is out of scope.
"""
import torch
from torch import nn


class RelationProjection(nn.Module):
    """Wrap a GNN model for relation projection."""

    def __init__(self, model, threshold=0.0):
        super(RelationProjection, self).__init__()
        self.model = model
        self.threshold = threshold

    def forward(self, graph, h_prob, r_index):
        bs = r_index.shape[0]
        # relation representations conditioned on the query relations, (bs, num_rel, dim)  (ultraquery.py:258)
        rel_reprs = self.model.relation_model(graph.relation_graph, query=r_index)
        query = rel_reprs[torch.arange(bs, device=r_index.device), r_index]            # (bs, dim)
        # initial node features: the fuzzy set scaled query vector (ultraquery.py:262); scores at or below the threshold
        # are cut off first (ultraquery.py:266-270: alleviates multi-source propagation)
        prob = h_prob
        if self.threshold > 0.0:
            prob = torch.where(h_prob <= self.threshold, torch.zeros_like(h_prob), h_prob)
        input = prob.unsqueeze(-1) * query.unsqueeze(1)                                # einsum("bn, bd -> bnd")
        output = self.model.entity_model(graph, input, rel_reprs, query)               # (bs, num_nodes) scores
        return torch.sigmoid(output)
