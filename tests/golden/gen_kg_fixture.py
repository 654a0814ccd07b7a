"""Writes tests/golden/kg_fixture/: a small knowledge graph in the raw layout of kg-datasets/FB15k-237 (PyG RelLinkPredDataset,
ultra/datasets.py:186-205) -- train.txt / valid.txt / test.txt with one tab-separated `head relation tail` triple per line,
entities.dict / relations.dict with `id name` per line -- from the synthetic generator, so that the reader
(ultra_amd.data.load_triples_dir) can be checked against the in-memory graph the triples came from.

    python tests/golden/gen_kg_fixture.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from ultra_amd import synthetic  # noqa: E402

NUM_NODE, NUM_TRAIN, NUM_REL, NUM_VALID, NUM_TEST, SEED = 300, 2400, 7, 40, 64, 11


def main():
    out = os.path.join(HERE, "kg_fixture")
    os.makedirs(out, exist_ok=True)
    data = synthetic.make_kg(num_node=NUM_NODE, num_triple=NUM_TRAIN, num_relation_base=NUM_REL, num_test=NUM_VALID + NUM_TEST,
                             seed=SEED, relation_graph=False)
    ent = ["/m/%04x" % (7919 * i % 65536) for i in range(NUM_NODE)]
    rel = ["/rel/r%d" % i for i in range(NUM_REL)]
    train = torch.stack([data.edge_index[0, :NUM_TRAIN], data.edge_index[1, :NUM_TRAIN], data.edge_type[:NUM_TRAIN]], dim=-1)
    held = data.target_triples
    splits = {"train.txt": train, "valid.txt": held[:NUM_VALID], "test.txt": held[NUM_VALID:]}
    for name, rows in splits.items():
        with open(os.path.join(out, name), "w") as f:
            for h, t, r in rows.tolist():
                f.write("%s\t%s\t%s\n" % (ent[h], rel[r], ent[t]))
    with open(os.path.join(out, "entities.dict"), "w") as f:
        for i, e in enumerate(ent):
            f.write("%d\t%s\n" % (i, e))
    with open(os.path.join(out, "relations.dict"), "w") as f:
        for i, r in enumerate(rel):
            f.write("%d\t%s\n" % (i, r))
    print("wrote", out, {k: len(v) for k, v in splits.items()})


if __name__ == "__main__":
    main()
