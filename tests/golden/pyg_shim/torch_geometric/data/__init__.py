import copy
import torch


class Data(object):
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def num_edges(self):
        return self.edge_index.shape[1]

    def __copy__(self):
        out = self.__class__.__new__(self.__class__)
        out.__dict__.update(self.__dict__)
        return out
