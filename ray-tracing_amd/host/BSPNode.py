"""Host mirror of BSPNode (BSPNode.py:8-210).

`build_bsp_tree` restates the reference's spatial-median split rule exactly
(BSPNode.py:54-121) because leaf order and child order define the tie-breaks of
the nearest-hit query; it works on index arrays instead of Python lists so that
million-primitive scenes build in seconds, and it records the tree directly in
the flat DFS pre-order layout the HIP library consumes (include/rt_amd.h).
`intersect_vectorize` keeps the reference's signature and is served by the GPU.
"""
import numpy as np

from surfaces.Object3D import Object3D
from util import EPSILON

LEAF_SIZE = 10  # BSPNode.py:67


class SurfaceContainer:
    def __init__(self, surface: Object3D):
        self.surface = surface
        self.intersection_indicator = 0


class FlatTree:
    """DFS pre-order arrays: node i's left child is i+1, `right[i]` its right child (-1 = leaf)."""

    def __init__(self):
        self.lo, self.hi, self.right, self.first, self.count = [], [], [], [], []
        self.leaf_prim = []   # chunks of finite-object positions (into the builder's `finite` list)
        self._n_refs = 0

    def finish(self):
        n = len(self.right)
        self.lo = np.asarray(self.lo, dtype=np.float64).reshape(n, 3)
        self.hi = np.asarray(self.hi, dtype=np.float64).reshape(n, 3)
        self.right = np.asarray(self.right, dtype=np.int32)
        self.first = np.asarray(self.first, dtype=np.int32)
        self.count = np.asarray(self.count, dtype=np.int32)
        self.leaf_prim = (np.concatenate(self.leaf_prim) if self.leaf_prim else np.zeros(0)).astype(np.int32)
        return self


class BSPNode:
    BSP_COUNTER = 0

    def __init__(self, surfaces=None, left=None, right=None, cut=None, small_bound=None, big_bound=None,
                 inf_plains=None):
        self._left = left
        self._right = right
        self._flat_index = None   # set on nodes that stand for a row of a GPU-built flat tree
        self.small_bound = small_bound
        self.big_bound = big_bound
        self.cut = cut
        self.inf_plains = inf_plains
        self._members = None      # positions into the root's finite-object list (leaves only)
        self._surfaces = surfaces
        self._root = self
        # root only:
        self.flat = None
        self.finite_objects = None
        self.all_surfaces = None
        self._ctx = None

    # children of a GPU-built tree are materialised from the flat arrays when somebody walks the objects
    def _child(self, which):
        i = self._flat_index
        flat = self._root.flat
        if i is None or flat.right[i] < 0:
            return None
        k = i + 1 if which == 0 else int(flat.right[i])
        return BSPNode._from_flat(self._root, k)

    @property
    def left(self):
        if self._left is None and self._flat_index is not None:
            self._left = self._child(0)
        return self._left

    @left.setter
    def left(self, node):
        self._left = node

    @property
    def right(self):
        if self._right is None and self._flat_index is not None:
            self._right = self._child(1)
        return self._right

    @right.setter
    def right(self, node):
        self._right = node

    @staticmethod
    def _from_flat(root, i):
        flat = root.flat
        node = BSPNode(small_bound=flat.lo[i], big_bound=flat.hi[i])
        node._root = root
        node._flat_index = i
        if flat.right[i] < 0:
            node._members = flat.leaf_prim[flat.first[i]:flat.first[i] + flat.count[i]]
        else:
            axis = int(np.argmax(flat.hi[i] - flat.lo[i]))          # BSPNode.py:71
            node.cut = ("xyz"[axis], flat.hi[i + 1][axis])           # the left child's big bound is the cut
        return node

    # leaves expose `surfaces` as a list of SurfaceContainer like the reference (built on demand)
    @property
    def surfaces(self):
        if self._surfaces is None and self._members is not None:
            fin = self._root.finite_objects
            self._surfaces = [SurfaceContainer(fin[k]) for k in self._members]
        return self._surfaces

    def __repr__(self, depth=0):
        pad = "  " * depth
        n = 0 if self._members is None else len(self._members)
        text = "%sBSPNode(cut=%s, bounds=(%s, %s), %d surfaces)\n" % (pad, self.cut, self.small_bound,
                                                                       self.big_bound, n)
        for child in (self.left, self.right):
            if child is not None:
                text += child.__repr__(depth + 1)
        return text

    @staticmethod
    def _corners(surfaces):
        finite = [o for o in surfaces if isinstance(o, Object3D)]
        planes = [o for o in surfaces if not isinstance(o, Object3D)]
        corners = [o.get_enclosing_values() for o in finite]
        small = np.array([c[0] for c in corners], dtype=np.float64).reshape(len(finite), 3)
        big = np.array([c[1] for c in corners], dtype=np.float64).reshape(len(finite), 3)
        return finite, planes, small, big

    @staticmethod
    def build_bsp_tree_device(surfaces, min_small_bound=None, max_big_bound=None, device=0):
        """The same tree as build_bsp_tree, built on the GPU by rt_bsp_build (csrc/rt_build.hip):
        identical flat arrays, hence identical nearest-hit tie-breaks.  Raises without a GPU."""
        from ray_tracing_amd import native
        finite, planes, small, big = BSPNode._corners(surfaces)
        t = native.bsp_build(small, big, min_small_bound, max_big_bound, LEAF_SIZE, EPSILON, device)
        flat = FlatTree()
        flat.lo, flat.hi, flat.right, flat.first, flat.count = t["lo"], t["hi"], t["right"], t["first"], t["count"]
        flat.leaf_prim = t["leaf_prim"]
        root = BSPNode(small_bound=flat.lo[0], big_bound=flat.hi[0], inf_plains=planes)
        root.flat = flat
        root._flat_index = 0
        if flat.right[0] < 0:
            root._members = flat.leaf_prim[:flat.count[0]]
        else:
            root.cut = BSPNode._from_flat(root, 0).cut
        root.finite_objects = finite
        root.all_surfaces = list(surfaces)
        root.build_stats = t["stats"]
        return root

    @staticmethod
    def build_bsp_tree(surfaces, min_small_bound=None, max_big_bound=None):
        finite, planes, small, big = BSPNode._corners(surfaces)
        if min_small_bound is None:
            min_small_bound = np.min(small, axis=0)
        if max_big_bound is None:
            max_big_bound = np.max(big, axis=0)
        flat = FlatTree()
        holder = []

        def grow(members, lo, hi, root_planes=None):
            """members: positions into `finite` in original order (BSPNode.py:55-97)."""
            me = len(flat.right)
            flat.lo.append(lo)
            flat.hi.append(hi)
            flat.right.append(-1)
            flat.first.append(-1)
            flat.count.append(0)
            extent = hi - lo
            if len(members) <= LEAF_SIZE or np.linalg.norm(extent) < EPSILON:
                node = BSPNode(small_bound=lo, big_bound=hi, inf_plains=root_planes)
                node._members = members
                flat.first[me] = flat._n_refs
                flat.count[me] = len(members)
                flat.leaf_prim.append(members)
                flat._n_refs += len(members)
                holder.append(node)
                return node
            axis = int(np.argmax(extent))
            where = (hi[axis] + lo[axis]) / 2
            top = big[members, axis]
            bottom = small[members, axis]
            left_only = top <= where
            right_only = ~left_only & (bottom >= where)
            left_members = members[~right_only]      # left-only and straddling, original order kept
            right_members = members[~left_only]      # right-only and straddling
            left_hi = np.array(hi, dtype=np.float64)
            left_hi[axis] = where
            right_lo = np.array(lo, dtype=np.float64)
            right_lo[axis] = where
            node = BSPNode(cut=("xyz"[axis], where), small_bound=lo, big_bound=hi, inf_plains=root_planes)
            node.left = grow(left_members, lo, left_hi)
            flat.right[me] = len(flat.right)
            node.right = grow(right_members, right_lo, hi)
            holder.append(node)
            return node

        import sys
        old = sys.getrecursionlimit()
        sys.setrecursionlimit(max(old, 10000))
        try:
            root = grow(np.arange(len(finite), dtype=np.int64), np.asarray(min_small_bound, dtype=np.float64),
                        np.asarray(max_big_bound, dtype=np.float64), planes)
        finally:
            sys.setrecursionlimit(old)
        for n in holder:
            n._root = root
        root.flat = flat.finish()
        root.finite_objects = finite
        root.all_surfaces = list(surfaces)
        return root

    # ------------------------------------------------------------------ queries (GPU)
    def _context(self):
        root = self._root
        if root._ctx is None:
            from ray_tracing_amd import native, scene_pack
            root._ctx = native.Context(scene_pack.pack_scene(root.all_surfaces, bsp_tree=root))
        return root._ctx

    def intersect_vectorize(self, ray_sources, ray_directions):
        """Nearest hit of every ray: (N,3) points and (N,) int64 surface indices, 0 = miss
        (BSPNode.py:201-210; a miss carries the Background point (1e8,1e8,1e8))."""
        if self is not self._root:
            raise NotImplementedError("intersect_vectorize is served for the root of the tree")
        BSPNode.BSP_COUNTER += 1
        return self._context().intersect(ray_sources, ray_directions)

    def direct_light_mask(self, ray_sources, ray_directions, targets):
        """The any-hit form of the query, as Light.compute_shadows uses it (Light.py:214-221): (N,) bool,
        True where the nearest hit of ray i lies within EPSILON of targets[i] - the same verdicts as
        `norm(intersect_vectorize(...)[0] - targets, axis=-1) < EPSILON`, without finishing the walks
        of occluded rays."""
        if self is not self._root:
            raise NotImplementedError("direct_light_mask is served for the root of the tree")
        BSPNode.BSP_COUNTER += 1
        return self._context().visibility(ray_sources, ray_directions, targets)
