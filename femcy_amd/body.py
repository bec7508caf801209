"""mesh container + topology, call-compatible with the reference's `Body`
(/root/reference/body.py:12-234) minus the Taichi GGUI rendering (out of scope: needs a display).

Topology queries that the reference builds with Python loops over all elements
(get_nodeEles :165-179, get_coElement_nodes :182-194, get_boundary :197-234) are vectorised
here; the device-side equivalents (node->element lists, adjacency, slot maps) are built natively
by femcy_build_pattern and never pass through Python.
"""
import numpy as np

from .fields import HostField


class Body:
    def __init__(self, nodes: np.ndarray, elements: np.ndarray, ELE) -> None:
        self.np_nodes = np.ascontiguousarray(nodes, dtype=np.float64)
        self.np_elements = np.ascontiguousarray(elements, dtype=np.int64)
        self.nodes = HostField(self.np_nodes)
        self.elements = HostField(self.np_elements, dtype=np.int32)
        self.dm = self.np_nodes.shape[1]
        self.ELE = ELE

    # -------------------------------------------------------------------------- topology
    def get_nodeEles(self, redo=False):
        """list (per node) of the elements that contain it, ascending."""
        if not hasattr(self, "nodeEles") or redo:
            el = self.np_elements
            order = np.argsort(el.ravel(), kind="stable")
            counts = np.bincount(el.ravel(), minlength=self.np_nodes.shape[0])
            owners = order // el.shape[1]
            self.nodeEles = [sorted(set(c.tolist())) for c in np.split(owners, np.cumsum(counts)[:-1])]
        return self.nodeEles

    def get_coElement_nodes(self, redo=False):
        """list (per node) of all nodes sharing an element with it (itself included), ascending."""
        if not hasattr(self, "coElement_nodes") or redo:
            import scipy.sparse as sp
            el = self.np_elements
            npe = el.shape[1]
            a = np.repeat(el, npe, axis=1).ravel()
            b = np.tile(el, (1, npe)).ravel()
            nn = self.np_nodes.shape[0]
            adj = sp.coo_matrix((np.ones(a.size, dtype=np.int8), (a, b)), shape=(nn, nn)).tocsr()
            adj.sum_duplicates()
            adj.sort_indices()
            self.coElement_nodes = [c.tolist() for c in np.split(adj.indices, adj.indptr[1:-1])]
        return self.coElement_nodes

    def get_boundary(self, redo=False):
        """facet (sorted global node tuple, facets as listed by ELE.facet_natural_coos) -> the one
        element that owns it, for facets that belong to a single element."""
        if not hasattr(self, "boundary") or redo:
            el = self.np_elements
            facets = list(self.ELE.facet_natural_coos.keys())
            sizes = {len(f) for f in facets}
            boundary = {}
            for k in sizes:                                   # all facets of one element type have one size; be general
                fk = [f for f in facets if len(f) == k]
                keys = np.concatenate([np.sort(el[:, list(f)], axis=1) for f in fk])          # [len(fk)*ne, k]
                owner = np.tile(np.arange(el.shape[0]), len(fk))
                order = np.lexsort(keys.T[::-1])
                ks = keys[order]
                new = np.ones(ks.shape[0], dtype=bool)
                new[1:] = (ks[1:] != ks[:-1]).any(axis=1)
                start = np.nonzero(new)[0]
                count = np.diff(np.append(start, ks.shape[0]))
                single = start[count == 1]
                boundary.update(zip(map(tuple, ks[single].tolist()), owner[order[single]].tolist()))
            self.boundary = boundary
            node2boundary = {}
            for f in self.boundary:
                for node in f:
                    node2boundary.setdefault(node, set()).add(f)
            self.node2boundary = node2boundary
            self.boundaryNodes = set(node2boundary.keys())
        return self.boundary

    @property
    def facetDic(self):
        """every facet -> list of the elements that hold it (reference body.py:203-216); built on demand, the
        boundary query above does not need the interior facets as Python objects."""
        if not hasattr(self, "_facetDic"):
            d = {}
            for facet in self.ELE.facet_natural_coos.keys():
                keys = np.sort(self.np_elements[:, list(facet)], axis=1)
                for iele, key in enumerate(map(tuple, keys.tolist())):
                    d.setdefault(key, []).append(iele)
            self._facetDic = d
        return self._facetDic

    def get_surfaceEdges(self, redo=False):
        if not hasattr(self, "surfaceEdges") or redo:
            pairs = np.concatenate([self.np_elements[:, [f[0], f[1]]] for f in self.ELE.facet_natural_coos.keys()])
            self.surfaceEdges = np.unique(np.sort(pairs, axis=1), axis=0)
        return self.surfaceEdges

    # ------------------------------------------------------------------------- rendering
    def show(self, *args, **kwargs):
        raise NotImplementedError("GGUI rendering is out of scope of femcy_amd (headless solve path); "
                                  "use system.dof.to_numpy() / system.cauchy_stress.to_numpy()")

    show2d = show
