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

Encodings 2 and 3 of the reference (adjacency-based elementary droplets) are not built.
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
