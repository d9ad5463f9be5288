"""Low-energy-spectrum bookkeeping for the branch-and-bound search: host-side only, it consumes what
the beam kernels already produce.

When branches with equal boundary indices are merged (tnac4o.py:481-509) the losers are not forgotten: each one differs
from the group's representative on a set of cells -- a *droplet* -- and costs `dE` more energy.  The reference keeps,
per surviving branch, a forest of such excitations (`el`), nested so that a droplet's own sub-droplets are only valid
together with it, and decodes the forest into explicit low-energy states afterwards.  This module restates the
"encoding 1" variant (independence of droplets decided by their extent along the row-major snake,
tnac4o.py:727-915, 2051-2079, 2249-2335, 1360-1389) with its own data structures:

  * `ShapeTable`: droplet shapes (cells that differ, xor of the cell states there) interned by content;
  * `ExcitationRecorder`: called once per site-step by the solver's merge, builds the new per-branch forests;
  * `unpack_snake`: enumerates the states encoded by a forest (energy above the ground state + list of shape ids).

A forest node is the tuple ((dE, shape_id, first_cell, last_cell, dlog2P), (children...)) -- the layout the reference
stores in its result files (`el`), so files written by either side decode on the other.

Encodings 2 and 3 decide independence from the interaction graph instead (tnac4o.py:943-1358, 2081-2247, 2337-2377):
only single-connected ("elementary") droplets are recorded, a droplet's sub-droplets are the loser's excitations that
touch it, and two excitations may be combined when they do not touch.  `Connectivity` answers the graph questions,
`AdjacencyRecorder` (encoding 2, nested) and `FlatRecorder` (encoding 3, one layer: every combination with the touching
sub-droplets is stored as its own merged shape) build the forests, `unpack_adjacent` enumerates them.  Nodes of these
encodings are ((dE, shape_id), (children...)).
"""
import numpy as np


def hamming_weight(dstate, mode):
    """Spins flipped by a droplet (tnac4o.py:2143-2150): Ising droplets are counted per differing cell entry as the
    reference does, RMF ones per set bit of the xor."""
    if mode == 'Ising':
        return len(dstate)
    return int(sum(bin(int(x)).count('1') for x in dstate))


class ShapeTable:
    """Interned droplet shapes: id -> (cells, xor values); equal shapes share one id (tnac4o.py:2051-2069)."""

    def __init__(self):
        self.by_id = {}
        self._by_content = {}
        self.next_id = 0

    @staticmethod
    def _key(cells, xors):
        return (np.asarray(cells, dtype=np.int64).tobytes(), np.asarray(xors, dtype=np.int64).tobytes())

    def intern(self, cells, xors):
        k = self._key(cells, xors)
        sid = self._by_content.get(k)
        if sid is None:
            sid = self.next_id
            self.next_id += 1
            self._by_content[k] = sid
            self.by_id[sid] = (cells, xors)
        return sid

    def keep_only(self, ids):
        """Drop every shape that no forest refers to any more (tnac4o.py:2249-2268)."""
        ids = set(ids)
        self.by_id = {i: v for i, v in self.by_id.items() if i in ids}
        self._by_content = {self._key(*v): i for i, v in self.by_id.items()}

    def semi_hash_index(self):
        """The reference's auxiliary index (first cell, first xor, last cell, last xor) -> ids, for its file format."""
        out = {}
        for i, (cells, xors) in self.by_id.items():
            out.setdefault((cells[0], xors[0], cells[-1], xors[-1]), []).append(i)
        return out


def prune(node, budget):
    """The node with every descendant chain that would exceed `budget` extra energy removed (tnac4o.py:2071-2079)."""
    head, children = node
    return (head, tuple(prune(c, budget - c[0][0]) for c in children if c[0][0] <= budget))


def shape_ids(forest, acc=None):
    acc = set() if acc is None else acc
    for head, children in forest:
        acc.add(head[1])
        shape_ids(children, acc)
    return acc


class ExcitationRecorder:
    """Per-branch excitation forests during the search.  `forests[k]` belongs to branch k of the current beam."""

    def __init__(self, max_dEng, lim_hd, mode):
        self.max_dEng, self.lim_hd, self.mode = max_dEng, lim_hd, mode
        self.shapes = ShapeTable()
        self.forests = [[]]

    def merge_step(self, site, parents, order, starts, Eng, prob, states, rep, probn, selected):
        """One site-step's merge.

        parents[i]: previous-beam branch the candidate i descends from; order / starts: candidates sorted by boundary
        group and the first position of each group in that order; Eng, prob, states: per candidate; rep[g]: the
        candidate that represents group g (lowest energy); probn[g]: the merged log2-probability of group g;
        selected: the groups that survive the top-M cut, in the order of the new beam (tnac4o.py:843-873)."""
        ends = np.r_[starts[1:], len(order)]
        new = []
        for g in selected:
            r = rep[g]
            forest = list(self.forests[parents[r]])
            for i in order[starts[g]:ends[g]]:
                dE = Eng[i] - Eng[r]
                if i == r or not (dE <= self.max_dEng):
                    continue
                diff = np.bitwise_xor(states[r], states[i])
                cells = diff.nonzero()[0]
                diff = diff[cells]
                if self.lim_hd > 1 and hamming_weight(diff, self.mode) < self.lim_hd:
                    continue
                first = cells[0]
                sid = self.shapes.intern(cells, diff)
                # the loser's own excitations survive inside the droplet if they start inside it and still fit the budget
                inner = tuple(prune(e, self.max_dEng - (e[0][0] + dE)) for e in self.forests[parents[i]]
                              if e[0][3] >= first and e[0][0] + dE <= self.max_dEng)
                forest.append(((dE, sid, first, site, prob[i] - probn[g]), inner))
            new.append(forest)
        self.forests = new
        used = set()
        for f in new:
            shape_ids(f, used)
        self.shapes.keep_only(used)

    def finish(self, order_i):
        """Forest of the best branch and the shape table with cell positions mapped back to the unrotated lattice
        (tnac4o.py:905-915)."""
        d = {}
        for sid, (cells, xors) in self.shapes.by_id.items():
            c = np.asarray(order_i)[cells]
            o = c.argsort()
            d[sid] = (c[o], xors[o])
        return self.forests[0], d


def unpack_snake(forest, ncells, max_dEng=0.0, max_states=np.inf):
    """All states encoded by `forest` with excitation energy <= max_dEng (tnac4o.py:2295-2335): returns (energies above
    the ground state, list of shape-id lists to flip).

    Cells are visited from the last one backwards.  A partial state carries a stack of open droplets (the innermost one
    offers its children); a child whose last cell is the current one may be added -- opening it -- and a droplet is
    closed once the walk has passed its first cell, which is what makes droplets that do not overlap along the snake
    independent.  When more than max_states partial states exist the highest-energy ones are dropped."""
    root = ((0.0, 0, -1, ncells - 1, 1.0), tuple(forest))
    energies, flips, stacks = [0.0], [[]], [[root]]
    for cell in range(ncells - 1, -1, -1):
        k = 0
        while k < len(energies):
            for child in stacks[k][-1][1]:
                head = child[0]
                if head[3] == cell and energies[k] + head[0] <= max_dEng:
                    energies.append(energies[k] + head[0])
                    flips.append(flips[k] + [head[1]])
                    stacks.append(stacks[k] + [child])
                elif head[3] > cell:
                    break
            k += 1
        if len(energies) > max_states:
            keep = np.array(energies).argpartition(max_states)[:max_states]
            energies = [energies[i] for i in keep]
            flips = [flips[i] for i in keep]
            stacks = [stacks[i] for i in keep]
        for st in stacks:
            while st[-1][0][2] >= cell:
                st.pop()
    return np.array(energies), flips


def decode_states(ground, forest, shapes, ncells, max_dEng, max_states, dtype):
    """Explicit low-energy states: the ground configuration with the droplets of each unpacked combination flipped
    (tnac4o.py:1360-1389).  Returns (excitation energies sorted ascending, states)."""
    E, flips = unpack_snake(forest, ncells, max_dEng=max_dEng, max_states=max_states)
    order = E.argsort()
    E = E[order]
    n = min(max_states, len(E))
    out = np.zeros((n, len(ground)), dtype=dtype)
    for i in range(n):
        st = ground.copy()
        for sid in flips[order[i]]:
            cells, xors = shapes[sid]
            st[cells] = np.bitwise_xor(st[cells], xors)
        out[i] = st
    return E, out


# ------------------------------------------------------------------------------------------------ encodings 2 and 3
class Connectivity:
    """Which droplets touch.  Ising: through the coupling graph of the spins they flip (adjacency from the nonzero
    off-diagonal couplings; a cell's xor value selects the flipped spins of that cell).  RMF: cells are nearest
    neighbours on the Nx-wide grid (tnac4o.py:2021-2041, 2081-2141)."""

    def __init__(self, mode, Nx, J=None, ind=None, adj=None):
        self.mode, self.Nx = mode, Nx
        if mode == 'Ising':
            if adj is None:
                up = np.triu(np.asarray(J), 1) != 0
                adj = up | up.T
            self.adj = np.asarray(adj, dtype=bool)
            self.flipped = []                      # per cell: xor value -> global spin indices
            for row in ind:
                for cell_spins in row:
                    cell_spins = np.asarray(cell_spins)
                    n = len(cell_spins)
                    masks = ((np.arange(2 ** n)[:, None] >> np.arange(n)[None, :]) & 1).astype(bool)
                    self.flipped.append([cell_spins[m] for m in masks])

    def spins(self, shape):
        cells, xors = shape
        return np.hstack([self.flipped[c][int(x) % len(self.flipped[c])] for c, x in zip(cells, xors)])

    def _grid_distance(self, a, b):
        ax, ay, bx, by = np.mod(a, self.Nx), a // self.Nx, np.mod(b, self.Nx), b // self.Nx
        return np.abs(ax[:, None] - bx[None, :]) + np.abs(ay[:, None] - by[None, :])

    def elementary(self, shape):
        """Single-connected?  Grown breadth-first from the first flipped spin / cell (tnac4o.py:2087-2114)."""
        if self.mode == 'Ising':
            nodes = self.spins(shape)
            front, rest = nodes[:1], nodes[1:]
            while front.size and rest.size:
                hit = np.any(self.adj[front][:, rest], axis=0)
                front, rest = rest[hit], rest[~hit]
            return rest.size == 0
        cells = np.asarray(shape[0])
        front, rest = cells[:1], cells[1:]
        while front.size and rest.size:
            hit = np.any(self._grid_distance(front, rest) == 1, axis=0)
            front, rest = rest[hit], rest[~hit]
        return rest.size == 0

    def touch(self, s1, s2):
        """tnac4o.py:2116-2141: coupled spins (Ising); same or neighbouring cells (RMF)."""
        if self.mode == 'Ising':
            return bool(np.any(self.adj[self.spins(s1)][:, self.spins(s2)]))
        return bool(np.any(self._grid_distance(np.asarray(s1[0]), np.asarray(s2[0])) <= 1))


def combine_shapes(s1, s2):
    """Composition of two droplets: xor on shared cells (cells that cancel drop out), union elsewhere (tnac4o.py:2198-2247)."""
    acc = {}
    for cells, xors in (s1, s2):
        for c, x in zip(cells, xors):
            c, x = int(c), int(x)
            acc[c] = acc[c] ^ x if c in acc else x
    cells = np.array(sorted(c for c, x in acc.items() if x != 0), dtype=np.int64)
    return cells, np.array([acc[c] for c in cells], dtype=np.int64)


def shape_distance(s1, s2, mode):
    """Hamming distance between the states two droplets lead to (tnac4o.py:2152-2196): Ising counts spins (bits of the
    xor), RMF counts cells."""
    a = {int(c): int(x) for c, x in zip(*s1)}
    b = {int(c): int(x) for c, x in zip(*s2)}
    hd = 0
    for c in set(a) | set(b):
        if mode == 'Ising':
            x = (a.get(c, 0) ^ b.get(c, 0)) if (c in a and c in b) else a.get(c, b.get(c))
            hd += bin(x).count('1')
        else:
            hd += 0 if (c in a and c in b and a[c] == b[c]) else 1
    return hd


def unpack_adjacent(forest, shapes, conn, max_dEng=0.0, max_states=np.inf, one_layer=False):
    """States encoded by a forest of encoding 2 (nested) or 3 (one_layer): (energies, shape-id lists).

    Every partial state owns a work list; taking its last excitation spawns a new state whose work list keeps only the
    excitations that do not touch the taken one, plus (nested encoding) the taken one's children.  The sweep over the
    states repeats while the previous one spawned something -- the reference's termination rule, kept as it is because
    it decides which states are listed (tnac4o.py:2337-2377)."""
    energies, flips, work = [0.0], [[]], [list(forest)]
    again = True
    while again:
        again = False
        k = 0
        while k < len(energies):
            if work[k]:
                node = work[k].pop()
                (dE, sid), children = node[0][:2], node[1]
                if energies[k] + dE <= max_dEng:
                    energies.append(energies[k] + dE)
                    flips.append(flips[k] + [sid])
                    left = [x for x in work[k] if not conn.touch(_shape(shapes, x[0][1]), _shape(shapes, sid))]
                    work.append(left)
                    if not one_layer:
                        left.extend(children)
                    if (not again) or left or work[k]:
                        again = True
            k += 1
        if len(energies) > max_states:
            keep = np.array(energies).argpartition(max_states)[:max_states]
            energies = [energies[i] for i in keep]
            flips = [flips[i] for i in keep]
            work = [work[i] for i in keep]
    return np.array(energies), flips


def _shape(shapes, ref):
    return shapes[ref] if isinstance(ref, (int, np.integer)) else ref


class AdjacencyRecorder(ExcitationRecorder):
    """Encoding 2: nested forests of elementary droplets, sub-droplets chosen by contact (tnac4o.py:1062-1088)."""

    def __init__(self, max_dEng, lim_hd, mode, conn):
        super().__init__(max_dEng, lim_hd, mode)
        self.conn = conn

    def _losers(self, g, order, starts, ends, Eng, states, rep):
        r = rep[g]
        for i in order[starts[g]:ends[g]]:
            dE = Eng[i] - Eng[r]
            if i == r or not (dE <= self.max_dEng):
                continue
            diff = np.bitwise_xor(states[r], states[i])
            cells = diff.nonzero()[0]
            yield i, dE, (cells, diff[cells])

    def merge_step(self, site, parents, order, starts, Eng, prob, states, rep, probn, selected):
        ends = np.r_[starts[1:], len(order)]
        new = []
        by_id = self.shapes.by_id
        for g in selected:
            forest = list(self.forests[parents[rep[g]]])
            for i, dE, shape in self._losers(g, order, starts, ends, Eng, states, rep):
                if self.lim_hd > 1 and hamming_weight(shape[1], self.mode) < self.lim_hd:
                    continue
                if not self.conn.elementary(shape):
                    continue
                sid = self.shapes.intern(*shape)
                inner = tuple(prune(e, self.max_dEng - (e[0][0] + dE)) for e in self.forests[parents[i]]
                              if e[0][0] + dE <= self.max_dEng and self.conn.touch(by_id[sid], by_id[e[0][1]]))
                forest.append(((dE, sid), inner))
            new.append(forest)
        self.forests = new
        self._drop_unused()

    def _drop_unused(self):
        used = set()
        for f in self.forests:
            shape_ids(f, used)
        self.shapes.keep_only(used)


class FlatRecorder(AdjacencyRecorder):
    """Encoding 3: one layer.  A loser's droplet is stored once per combination with its touching sub-droplets, each
    combination as its own merged elementary shape with the summed energy (tnac4o.py:1261-1285); the shape table is
    cleaned once per row, and near-duplicates (Hamming distance < lim_hd) are removed greedily at the end (:1324-1339)."""

    def merge_step(self, site, parents, order, starts, Eng, prob, states, rep, probn, selected):
        ends = np.r_[starts[1:], len(order)]
        new = []
        by_id = self.shapes.by_id
        for g in selected:
            forest = list(self.forests[parents[rep[g]]])
            fresh = []
            for i, dE, shape in self._losers(g, order, starts, ends, Eng, states, rep):
                near = [e for e in self.forests[parents[i]]
                        if e[0][0] + dE <= self.max_dEng and self.conn.touch(shape, by_id[e[0][1]])]
                sE, sflip = unpack_adjacent(near, by_id, self.conn, self.max_dEng - dE, one_layer=True)
                for e_sub, ids in zip(sE, sflip):
                    merged = shape
                    for sid in ids:
                        merged = combine_shapes(merged, by_id[sid])
                    if (self.lim_hd <= 1 or hamming_weight(merged[1], self.mode) >= self.lim_hd) and self.conn.elementary(merged):
                        fresh.append(((e_sub + dE, self.shapes.intern(*merged)), ()))
            fresh.sort(key=lambda node: node[0][0])
            forest.extend(fresh)
            new.append(forest)
        self.forests = new

    def end_row(self):
        self._drop_unused()

    def finish(self, order_i):
        best = sorted(self.forests[0], key=lambda node: node[0][0])
        if self.lim_hd > 1:
            kept = []
            for node in best:
                if all(shape_distance(self.shapes.by_id[node[0][1]], self.shapes.by_id[o[0][1]], self.mode) >= self.lim_hd
                       for o in kept):
                    kept.append(node)
            best = kept
        self.forests[0] = best
        self._drop_unused()
        return super().finish(order_i)


def decode_states_adjacent(ground, forest, shapes, conn, max_dEng, max_states, dtype, one_layer):
    """decode_states for encodings 2 / 3."""
    E, flips = unpack_adjacent(forest, shapes, conn, max_dEng=max_dEng, max_states=max_states, one_layer=one_layer)
    order = E.argsort()
    E = E[order]
    n = min(max_states, len(E))
    out = np.zeros((n, len(ground)), dtype=dtype)
    for i in range(n):
        st = ground.copy()
        for sid in flips[order[i]]:
            cells, xors = shapes[sid]
            st[cells] = np.bitwise_xor(st[cells], xors)
        out[i] = st
    return E, out
