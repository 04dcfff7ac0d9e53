"""Host-side (theta -> pi, T, E) construction for the pairwise CoalHMMs.

The forward engine consumes ``(pi, T, E)``; in the reference those come from the CPU model layer
(src/IMCoalHMM/model.py:44-49 -> transitions.py:204-248, emissions.py:89-100, CTMC.py:39-51, the
state spaces of state_spaces.py / statespace_generator.py).  Once a forward pass costs about a
millisecond that layer is the wall (0.05-1.6 s per HMM in the reference), so it is rebuilt here,
still on the CPU as north_star asks, around one generic engine:

* **state spaces** are enumerated once per process over integer-coded lineages
  ``(population, left nucleotides, right nucleotides)`` and stored as class index arrays
  (B = neither locus coalesced, L = left only, R = right only, E = both);
* every demographic model is described as a *piecewise CTMC*: a start vector at the first break
  point plus one ``through`` matrix per HMM interval (``expm`` of a rate matrix times the interval
  length, de-duplicated and evaluated as one stacked ``scipy.linalg.expm`` call), with a 0/1
  projection where the state space changes;
* the joint two-locus coalescence-interval matrix ``J`` (transitions.py:217-235) is accumulated
  by a vector recursion over the L class only - ``O(n^2 |L|)`` instead of the reference's
  ``O(n^2 |S|^3)`` table of ``between`` matrices - and ``pi = J 1``, ``T = J / pi`` follow
  (transitions.py:239-246).

The models keep the reference's constructor and method names (``valid_parameters``,
``emission_points``, ``build_hidden_markov_model``) so that ``Likelihood(model, forwarders)`` works
with either implementation; ``build_batch(thetas)`` is the additive population entry point.
Parity is pinned: tests/test_models_cpu.py compares against (pi, T, E) produced by the reference's
own model classes (tests/golden/model_golden.npz, generator committed beside it).
"""
import ctypes
import math
import os
from collections import deque

import numpy as np
from scipy.linalg import expm as _scipy_expm

__all__ = [
    "exp_break_points", "trunc_exp_break_points", "uniform_break_points", "psmc_break_points",
    "coalescence_points", "emission_matrix", "StateSpace", "isolation_space", "single_space",
    "migration_space", "IsolationModel", "IsolationMigrationModel",
    "VariableCoalescenceRateIsolationModel", "VariableCoalAndMigrationRateModel",
    "IsolationMigrationEpochsModel",
]


# ---------------------------------------------------------------------------------------------
# break points (break_points.py:9-107) and emissions (emissions.py:11-100)
# ---------------------------------------------------------------------------------------------

def exp_break_points(no_intervals, coal_rate, offset=0.0):
    """Equal-probability break points of Exp(coal_rate), shifted by ``offset`` (break_points.py:9-31)."""
    q = np.arange(no_intervals, dtype=np.float64) / no_intervals
    return -np.log1p(-q) / coal_rate + offset


def trunc_exp_break_points(no_intervals, coal_rate, end, offset=0.0):
    """Equal-probability break points of Exp(coal_rate) truncated at ``end`` (break_points.py:34-60)."""
    p = np.arange(no_intervals, dtype=np.float64) / no_intervals
    tail = math.exp(-coal_rate * end) - 1.0
    return -np.log(1.0 + tail * p) / coal_rate + offset


def uniform_break_points(no_intervals, start, end):
    """``no_intervals`` equidistant points from ``start`` (included) to ``end`` (excluded) (break_points.py:62-80)."""
    q = np.arange(no_intervals, dtype=np.float64) / no_intervals
    return q * (end - start) + start


def psmc_break_points(no_intervals=64, t_max=15, mu=1e-9, offset=0.0):
    """Li & Durbin (2011) spacing, as break_points.py:83-107 scales it."""
    scale = math.log(1 + 10 * t_max * mu)
    return [offset] + [offset + 0.1 * (math.exp(float(i) / no_intervals * scale) - 1.0)
                       for i in range(1, no_intervals)]


def coalescence_points(break_points, rates):
    """Mean coalescence time inside every interval and after the last break point (emissions.py:44-66)."""
    bp = np.asarray(break_points, dtype=np.float64)
    if np.ndim(rates) == 0:
        rates = np.full(len(bp), float(rates))
    else:
        rates = np.asarray(rates, dtype=np.float64)
        if len(rates) != len(bp):
            raise ValueError("You must have the same number of rates as break points.")
    out = np.empty(len(bp))
    dt = bp[1:] - bp[:-1]
    r = rates[:-1]
    decay = np.exp(-dt * r)
    out[:-1] = bp[:-1] + 1.0 / r - (dt * decay) / (1 - decay)       # truncated exponential mean
    out[-1] = bp[-1] + 1.0 / rates[-1]
    return out


def emission_matrix(coal_points):
    """Jukes-Cantor emissions at twice the coalescence time; column 2 (missing data) is 1 (emissions.py:89-100)."""
    t = np.asarray(coal_points, dtype=np.float64)
    decay = np.exp(-4.0 / 3 * (2 * t))
    E = np.empty((len(t), 3))
    E[:, 0] = 0.25 + 0.75 * decay
    E[:, 1] = 0.75 - 0.75 * decay
    E[:, 2] = 1.0
    return E


# ---------------------------------------------------------------------------------------------
# state spaces (statespace_generator.py:23-185, state_spaces.py:7-130)
# ---------------------------------------------------------------------------------------------
# A lineage is (population, left mask, right mask); masks are subsets of the two samples {1, 2}
# as bits 0 and 1.  It is packed as population*16 + left*4 + right; a state is the sorted tuple
# of its lineages.

def _pack(pop, left, right):
    return pop * 16 + left * 4 + right


def _unpack(code):
    return code >> 4, (code >> 2) & 3, code & 3


class StateSpace(object):
    """Reachable two-locus, two-sample ancestral configurations and their labelled transitions.

    ``edges`` is a list of ``(src, (kind, pop_from, pop_to), dst)`` with kind in ``"C"``
    (coalescence inside a population), ``"R"`` (recombination inside a population) and ``"M"``
    (migration pop_from -> pop_to), the label convention of state_spaces.py:31-39,66-71,120-130.
    States are numbered class by class: B, L, R, E.
    """

    def __init__(self, init, migrations=()):
        """``init``: iterable of lineages (pop, left, right); ``migrations``: allowed (from, to) moves."""
        start = tuple(sorted(_pack(*lin) for lin in init))
        moves = {}
        for a, b in migrations:
            moves.setdefault(a, []).append(b)
        seen = {start}
        todo = deque([start])
        raw_edges = []
        while todo:
            state = todo.popleft()
            for label, nxt in self._successors(state, moves):
                if nxt not in seen:
                    seen.add(nxt)
                    todo.append(nxt)
                raw_edges.append((state, label, nxt))

        def klass(state):
            left = any(((c >> 2) & 3) == 3 for c in state)
            right = any((c & 3) == 3 for c in state)
            return int(left) + 2 * int(right)

        ordered = sorted(seen, key=lambda s: (klass(s), s))
        self.index = {s: k for k, s in enumerate(ordered)}
        self.states = ordered
        self.size = len(ordered)
        kl = np.array([klass(s) for s in ordered])
        self.begin_states = np.flatnonzero(kl == 0)
        self.left_states = np.flatnonzero(kl == 1)
        self.right_states = np.flatnonzero(kl == 2)
        self.end_states = np.flatnonzero(kl == 3)
        self.edges = [(self.index[s], label, self.index[d]) for s, label, d in raw_edges]
        self._src = np.array([e[0] for e in self.edges])
        self._dst = np.array([e[2] for e in self.edges])
        self.labels = sorted(set(e[1] for e in self.edges))
        self._label_of_edge = np.array([self.labels.index(e[1]) for e in self.edges])

    @staticmethod
    def _successors(state, moves):
        for k, code in enumerate(state):
            pop, left, right = _unpack(code)
            rest = state[:k] + state[k + 1:]
            if left and right:     # recombination un-links the two loci (statespace_generator.py:159-171)
                yield ("R", pop, pop), tuple(sorted(rest + (_pack(pop, left, 0), _pack(pop, 0, right))))
            for other in moves.get(pop, ()):
                yield ("M", pop, other), tuple(sorted(rest + (_pack(other, left, right),)))
        for a in range(len(state)):
            for b in range(a):     # coalescence needs both lineages in one population (:173-185)
                pa, la, ra = _unpack(state[a])
                pb, lb, rb = _unpack(state[b])
                if pa != pb:
                    continue
                rest = tuple(c for k, c in enumerate(state) if k != a and k != b)
                yield ("C", pa, pa), tuple(sorted(rest + (_pack(pa, la | lb, ra | rb),)))

    def state_of(self, lineages):
        return self.index[tuple(sorted(_pack(*lin) for lin in lineages))]

    def rate_matrix(self, rates):
        """Generator matrix for a ``{label: rate}`` table (CTMC.py:12-37)."""
        per_label = np.array([float(rates[label]) for label in self.labels])
        flat = getattr(self, "_flat_edges", None)
        if flat is None:
            flat = self._flat_edges = self._src * self.size + self._dst
        # (bincount adds the weights of equal (src, dst) pairs in edge order, as np.add.at did - at a third of the time)
        Q = np.bincount(flat, weights=per_label[self._label_of_edge], minlength=self.size * self.size).reshape(self.size, self.size)
        Q.ravel()[::self.size + 1] = -Q.sum(axis=1)
        return Q

    def projection_to(self, other, merge_populations):
        """0/1 matrix sending every state to its image in ``other`` (transitions.py:11-32)."""
        P = np.zeros((self.size, other.size))
        for state, k in self.index.items():
            if merge_populations:
                image = tuple(sorted(c & 15 for c in state))
            else:
                image = state
            P[k, other.index[image]] = 1.0
        return P


_SPACES = {}


def isolation_space():
    """Two samples in two isolated populations (state_spaces.py:7-28): 4 states."""
    if "iso" not in _SPACES:
        sp = StateSpace([(1, 1, 1), (2, 2, 2)])
        sp.i12_index = sp.state_of([(1, 1, 1), (2, 2, 2)])
        _SPACES["iso"] = sp
    return _SPACES["iso"]


def single_space():
    """Two samples in one (ancestral) population (state_spaces.py:42-63): 15 states."""
    if "single" not in _SPACES:
        _SPACES["single"] = StateSpace([(0, 1, 1), (0, 2, 2)])
    return _SPACES["single"]


def migration_space():
    """Two populations exchanging migrants (state_spaces.py:74-117): 94 states."""
    if "mig" not in _SPACES:
        sp = StateSpace([(1, 1, 1), (2, 2, 2)], migrations=[(1, 2), (2, 1)])
        sp.i11_index = sp.state_of([(1, 1, 1), (1, 2, 2)])
        sp.i12_index = sp.state_of([(1, 1, 1), (2, 2, 2)])
        sp.i22_index = sp.state_of([(2, 1, 1), (2, 2, 2)])
        _SPACES["mig"] = sp
    return _SPACES["mig"]


def isolation_rates(coal_rate_1, coal_rate_2, recomb_rate):
    """state_spaces.py:31-39"""
    return {("C", 1, 1): coal_rate_1, ("C", 2, 2): coal_rate_2,
            ("R", 1, 1): recomb_rate, ("R", 2, 2): recomb_rate}


def single_rates(coal_rate, recomb_rate):
    """state_spaces.py:66-71"""
    return {("C", 0, 0): coal_rate, ("R", 0, 0): recomb_rate}


def migration_rates(coal_rate_1, coal_rate_2, recomb_rate, migration_rate_12, migration_rate_21):
    """state_spaces.py:120-130"""
    return {("C", 1, 1): coal_rate_1, ("C", 2, 2): coal_rate_2,
            ("R", 1, 1): recomb_rate, ("R", 2, 2): recomb_rate,
            ("M", 1, 2): migration_rate_12, ("M", 2, 1): migration_rate_21}


# ---------------------------------------------------------------------------------------------
# the generic engine: piecewise CTMC -> joint matrix -> (pi, T)
# ---------------------------------------------------------------------------------------------

class PiecewiseCTMC(object):
    """What every model hands to :func:`hmm_transitions`.

    ``start``    row vector over ``spaces[0]``: state distribution at the first break point
                 (``upto_[0][initial, :]`` of transitions.py:34-55);
    ``spaces``   the StateSpace of every HMM interval (``get_state_space(i)``);
    ``pieces``   per interval but the last: ``(Q, dt, projection or None)`` meaning
                 ``through_i = expm(Q dt) @ projection``; the last interval's pseudo matrix
                 (isolation_model.py:38-47) is implied.
    """

    def __init__(self, start, spaces, pieces):
        self.start = start
        self.spaces = spaces
        self.pieces = pieces
        assert len(pieces) == len(spaces) - 1


class _Through(object):
    """``through[i]`` of B systems without materialising ``(B, |S_i|, |S_i+1|)`` arrays.

    Each distinct ``(Q, dt)`` of a system is exponentiated once (CTMC.py:39-51 caches the same
    way); all matrices of one size - across intervals and across the B systems - go through a
    single stacked ``scipy.linalg.expm`` call, and the recursion below only gathers the class
    blocks it needs from those stacks.
    """

    def __init__(self, systems):
        slots = {}                 # (system, id(Q), dt) -> position in that size's stack
        members = {}
        for b, system in enumerate(systems):
            for Q, dt, _ in system.pieces:
                key = (b, id(Q), float(dt))
                if key not in slots:
                    group = members.setdefault(Q.shape[0], [])
                    slots[key] = len(group)
                    group.append(Q * dt)
        self.stacks = {size: _scipy_expm(np.stack(group)) for size, group in members.items()}
        self._gathered = {}
        self.where = []            # per interval: (size, positions (B,), projection)
        for i in range(len(systems[0].pieces)):
            pos = np.array([slots[(b, id(sy.pieces[i][0]), float(sy.pieces[i][1]))]
                            for b, sy in enumerate(systems)])
            self.where.append((systems[0].pieces[i][0].shape[0], pos, systems[0].pieces[i][2]))

    def block(self, i, rows, cols):
        """``through_i[rows, cols]`` for every system: ``(B, len(rows), len(cols))``."""
        size, pos, proj = self.where[i]
        stack = self.stacks[size]
        if proj is None:
            # one gather per (matrix size, row class, column class) for ALL matrices of the stack; an interval
            # then only picks its systems' positions (the class index arrays live on the StateSpace objects)
            key = (size, id(rows), id(cols))
            blocks = self._gathered.get(key)
            if blocks is None:
                blocks = self._gathered[key] = stack[:, rows[:, None], cols[None, :]]
            return blocks[pos]
        return stack[pos[:, None], rows[None, :]] @ proj[:, cols]


# ---------------------------------------------------------------------------------------------
# native path (csrc/model_host.hpp behind include/imcoal_model.h): the same recursion in C++ for
# small state spaces, where the ~40 numpy calls per interval - not the arithmetic - are the cost
# ---------------------------------------------------------------------------------------------
NATIVE_MAX_SPACE = 32           # larger spaces (the 94-state migration space) stay on numpy / BLAS
_native = {"lib": None, "tried": False, "structures": {}}


def _native_lib():
    """libimcoal_fwd.so if it is there (host-only entry points: no device is touched), else None -
    the numpy path computes the same numbers.  IMC_MODEL_NATIVE=0 switches the native path off."""
    if not _native["tried"]:
        _native["tried"] = True
        if os.environ.get("IMC_MODEL_NATIVE", "1") != "0":
            try:
                from . import _capi
                _native["lib"] = _capi.lib()
            except Exception:            # library not built: numpy path
                _native["lib"] = None
    return _native["lib"]


def expm(A):
    """``scipy.linalg.expm`` for stacks and large matrices; one small matrix goes through the
    library's [13/13] Pade routine (a 4 x 4 scipy call costs 60 us of Python)."""
    A = np.asarray(A, dtype=np.float64)
    lib = _native_lib() if A.ndim == 2 and A.shape[0] <= NATIVE_MAX_SPACE else None
    if lib is None:
        return _scipy_expm(A)
    A = np.ascontiguousarray(A)
    out = np.empty_like(A)
    call = _native.get("expm_call")
    if call is None:
        call = _native["expm_call"] = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)(("imc_model_expm", lib))
    if call(A.shape[0], A.ctypes.data, out.ctypes.data) != 0:
        return _scipy_expm(A)
    return out


def _native_structure(system):
    """The parameter-independent part of a PiecewiseCTMC as the int32 arrays of imc_model_transitions,
    cached per (state spaces, piece pattern, projections)."""
    q_index, piece_q, projs, piece_proj = {}, [], [], []
    for Q, _, proj in system.pieces:
        piece_q.append(q_index.setdefault(id(Q), len(q_index)))
        if proj is None:
            piece_proj.append(-1)
        else:
            for k, known in enumerate(projs):
                if known is proj:
                    piece_proj.append(k)
                    break
            else:
                projs.append(proj)
                piece_proj.append(len(projs) - 1)
    key = (tuple(id(sp) for sp in system.spaces), tuple(piece_q), tuple(piece_proj), tuple(id(p) for p in projs))
    st = _native["structures"].get(key)
    if st is None:
        spaces = system.spaces
        if max(sp.size for sp in spaces) > NATIVE_MAX_SPACE:
            st = False
        else:
            lists = [np.asarray(c, dtype=np.int32) for sp in spaces for c in (sp.begin_states, sp.left_states, sp.end_states)]
            proj_flat = [np.ascontiguousarray(p, dtype=np.float64).ravel() for p in projs]
            st = {
                "n": len(spaces),
                "space_size": np.array([sp.size for sp in spaces], dtype=np.int32),
                "cls_off": np.concatenate([[0], np.cumsum([len(c) for c in lists])]).astype(np.int32),
                "cls_idx": np.concatenate(lists).astype(np.int32) if lists else np.zeros(0, dtype=np.int32),
                "piece_q": np.array(piece_q, dtype=np.int32), "piece_proj": np.array(piece_proj, dtype=np.int32),
                "n_q": len(q_index),
                "proj_off": np.concatenate([[0], np.cumsum([len(p) for p in proj_flat])])[:-1].astype(np.int32) if projs else np.zeros(0, dtype=np.int32),
                "proj": np.concatenate(proj_flat) if projs else np.zeros(0),
                "keep": (list(spaces), projs),           # the ids in the key stay valid while the entry lives
            }
        _native["structures"][key] = st
    return st, q_index


def _native_transitions(systems):
    lib = _native_lib()
    if lib is None:
        return None
    if len(systems[0].spaces) < 2:          # a single interval: nothing to exponentiate, numpy does it
        return None
    st, q_index = _native_structure(systems[0])
    if not st:
        return None
    n, nb = st["n"], len(systems)
    order = sorted(q_index, key=q_index.get)
    first = {id(Q): Q for Q, _, _ in systems[0].pieces}
    q_size = np.array([first[i].shape[0] for i in order], dtype=np.int32)
    Qs = np.empty((nb, int((q_size.astype(np.int64) ** 2).sum())))
    dts = np.empty((nb, max(n - 1, 0)))
    s0 = int(st["space_size"][0])
    starts = np.empty((nb, s0))
    q_pos = np.concatenate([[0], np.cumsum(q_size.astype(np.int64) ** 2)])
    piece_q = st["piece_q"]
    for b, sy in enumerate(systems):
        if len(sy.pieces) != n - 1:
            return None
        seen = {}
        for i, (Q, dt, _) in enumerate(sy.pieces):
            k = seen.get(id(Q))
            if k is None:                                      # first appearance: copy the matrix
                k = seen[id(Q)] = len(seen)
                if k >= len(q_size) or Q.shape[0] != q_size[k]:
                    return None                                # another piece pattern: numpy path
                Qs[b, q_pos[k]:q_pos[k + 1]] = np.asarray(Q, dtype=np.float64).ravel()
            if k != piece_q[i]:
                return None
            dts[b, i] = dt
        starts[b] = sy.start
    pi = np.empty((nb, n))
    T = np.empty((nb, n, n))
    threads = min(16, os.cpu_count() or 1) if nb >= 8 else 1
    call = _native.get("call")
    if call is None:            # raw addresses instead of typed pointer objects: 15 data_as() calls cost more than the recursion
        vp = ctypes.c_void_p
        proto = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_int, vp, ctypes.c_int, vp, vp,
                                 vp, vp, vp, vp, vp, ctypes.c_int)
        call = _native["call"] = proto(("imc_model_transitions", lib))
    addr = st.get("addr")
    if addr is None:
        addr = st["addr"] = tuple(st[k].ctypes.data for k in ("space_size", "cls_off", "cls_idx", "piece_q", "piece_proj", "proj_off", "proj"))
    rc = call(nb, n, addr[0], addr[1], addr[2], addr[3], addr[4], st["n_q"], q_size.ctypes.data, len(st["proj_off"]), addr[5], addr[6],
              Qs.ctypes.data, dts.ctypes.data, starts.ctypes.data, pi.ctypes.data, T.ctypes.data, threads)
    if rc != 0:
        msg = lib.imc_last_error().decode("utf-8", "replace")
        if "must be supported on the B class" in msg:
            raise ValueError(msg)
        raise AssertionError(msg)
    return pi, T


def hmm_transitions_batch(systems):
    """Initial distributions ``(B, n)`` and transition matrices ``(B, n, n)`` of B CoalHMMs that
    share one interval structure (transitions.py:204-248).

    ``J[i, j]`` is the probability that the left locus coalesces in interval i and the right
    locus in interval j.  Only three blocks of every through matrix matter.  The chain starts in
    the B class (neither locus coalesced) and B is only entered from B, so the distribution at
    the start of interval i, restricted to B, obeys ``b_i = b_{i-1} through_{i-1}[B, B]``.
    The diagonal is ``b_i through_i[B, E] 1``; for i < j the row vector ``b_i through_i[B, L]``
    is carried forward through the L class only (a coalesced locus stays coalesced, so L -> L
    paths never leave L) and closed with ``through_j[L, E] 1``.  All B systems advance together
    as stacked matrix products.
    """
    done = _native_transitions(systems)
    if done is not None:
        return done
    spaces = systems[0].spaces
    n = len(spaces)
    nb = len(systems)
    through = _Through(systems)
    J = np.zeros((nb, n, n))

    begin = [None] * n             # begin[i]: (B, 1, |B_i|)
    start = np.stack([np.asarray(s.start, dtype=np.float64) for s in systems])
    outside = np.ones(start.shape[1], dtype=bool)
    outside[spaces[0].begin_states] = False
    if np.any(start[:, outside] != 0.0):
        raise ValueError("the start distribution must be supported on the B class")
    begin[0] = start[:, None, spaces[0].begin_states]
    for i in range(1, n):
        begin[i] = begin[i - 1] @ through.block(i - 1, spaces[i - 1].begin_states, spaces[i].begin_states)

    # diagonal (transitions.py:217-224); the i = 0 term applies interval 0's E indices to the
    # distribution over interval 1's space, exactly as the reference does
    for i in range(0, n - 1):
        ends = spaces[0].end_states if i == 0 else spaces[i + 1].end_states
        J[:, i, i] = (begin[i] @ through.block(i, spaces[i].begin_states, ends)).sum(axis=(1, 2))
    J[:, n - 1, n - 1] = begin[n - 1].sum(axis=(1, 2))

    # i < j (transitions.py:226-235): row i of V is interval i's vector, columns are the L class of interval j
    V = None
    for j in range(1, n):
        Lj = spaces[j].left_states
        if V is None:
            V = np.zeros((nb, n - 1, len(Lj)))
        V[:, j - 1, :] = (begin[j - 1] @ through.block(j - 1, spaces[j - 1].begin_states, Lj))[:, 0, :]
        if j == n - 1:
            J[:, :j, j] = V[:, :j].sum(axis=2)      # pseudo through matrix: every L state ends in E
            break
        closing = through.block(j, Lj, spaces[j + 1].end_states).sum(axis=2)
        J[:, :j, j] = (V[:, :j] @ closing[:, :, None])[:, :, 0]
        Lnext = spaces[j + 1].left_states
        carried = np.zeros((nb, n - 1, len(Lnext)))
        carried[:, :j] = V[:, :j] @ through.block(j, Lj, Lnext)
        V = carried
    upper = np.triu_indices(n, 1)
    J[:, upper[1], upper[0]] = J[:, upper[0], upper[1]]
    total = J.sum(axis=(1, 2))
    if not np.all(np.abs(total - 1.0) < 1.5e-7):  # numpy.testing.assert_almost_equal, 7 decimals (:237)
        raise AssertionError("joint probabilities sum to %r, not 1" % total)
    pi = J.sum(axis=2)
    T = J / pi[:, :, None]
    return pi, T


def hmm_transitions(system):
    """``(pi, T)`` of one CoalHMM; see :func:`hmm_transitions_batch`."""
    pi, T = hmm_transitions_batch([system])
    return pi[0], T[0]


# ---------------------------------------------------------------------------------------------
# models
# ---------------------------------------------------------------------------------------------

class Model(object):
    """Common surface of the demographic models (model.py:11-49)."""

    def valid_parameters(self, parameters):
        """All parameters positive (model.py:32-42)."""
        assert isinstance(parameters, np.ndarray)
        return bool(np.all(parameters > 0))

    def build_ctmc_system(self, *parameters):
        raise NotImplementedError

    def emission_points(self, *parameters):
        raise NotImplementedError

    def build_hidden_markov_model(self, parameters):
        """theta -> (pi (N,), T (N,N) row-stochastic, E (N,3)) (model.py:44-49)."""
        pi, T = hmm_transitions(self.build_ctmc_system(*parameters))
        E = emission_matrix(self.emission_points(*parameters))
        return pi, T, E

    def build_batch(self, thetas):
        """Stacked ``(pis[B,N], Ts[B,N,N], Es[B,N,3])`` for a population of parameter points:
        one stacked expm and one stacked recursion for the whole population."""
        thetas = [np.asarray(t, dtype=np.float64) for t in thetas]
        pis, Ts = hmm_transitions_batch([self.build_ctmc_system(*t) for t in thetas])
        Es = np.stack([emission_matrix(self.emission_points(*t)) for t in thetas])
        return pis, Ts, Es


def _repeat_per_epoch(values, intervals):
    if len(values) != len(intervals):
        raise ValueError("one value per epoch expected")
    return [v for v, count in zip(values, intervals) for _ in range(count)]


def _pieces_for(Qs, break_points):
    """(Q_i, t_{i+1} - t_i, None) for consecutive break points."""
    return [(Qs[i], break_points[i + 1] - break_points[i], None) for i in range(len(break_points) - 1)]


class IsolationModel(Model):
    """Clean split at ``split_time`` into one ancestral population (isolation_model.py:96-130).

    Parameters: ``(split_time, coal_rate, recomb_rate)``.
    """

    def __init__(self, no_hmm_states):
        self.no_hmm_states = no_hmm_states
        self.isolation_state_space = isolation_space()
        self.single_state_space = single_space()
        self._iso_to_single = self.isolation_state_space.projection_to(self.single_state_space, True)

    def emission_points(self, split_time, coal_rate, _):
        return coalescence_points(exp_break_points(self.no_hmm_states, coal_rate, split_time), coal_rate)

    def build_ctmc_system(self, split_time, coal_rate, recomb_rate):
        iso, single = self.isolation_state_space, self.single_state_space
        Qi = iso.rate_matrix(isolation_rates(coal_rate, coal_rate, recomb_rate))
        Qs = single.rate_matrix(single_rates(coal_rate, recomb_rate))
        bp = exp_break_points(self.no_hmm_states, coal_rate, split_time)
        start = expm(Qi * bp[0])[iso.i12_index] @ self._iso_to_single
        n = self.no_hmm_states
        return PiecewiseCTMC(start, [single] * n, _pieces_for([Qs] * n, bp))


class IsolationMigrationModel(Model):
    """Isolation, then a period of symmetric migration, then one ancestral population
    (isolation_with_migration_model.py:124-176).

    Parameters: ``(isolation_time, migration_time, coal_rate, recomb_rate, mig_rate)``.
    """

    def __init__(self, no_mig_states, no_ancestral_states):
        self.isolation_state_space = isolation_space()
        self.migration_state_space = migration_space()
        self.single_state_space = single_space()
        self.no_mig_states = no_mig_states
        self.no_ancestral_states = no_ancestral_states
        self._iso_to_mig = self.isolation_state_space.projection_to(self.migration_state_space, False)
        self._mig_to_single = self.migration_state_space.projection_to(self.single_state_space, True)

    def _break_points(self, isolation_time, migration_time, coal_rate):
        tau1 = isolation_time
        tau2 = isolation_time + migration_time
        return (uniform_break_points(self.no_mig_states, tau1, tau2),
                exp_break_points(self.no_ancestral_states, coal_rate, tau2))

    def emission_points(self, isolation_time, migration_time, coal_rate, recomb_rate, mig_rate):
        mig_bp, anc_bp = self._break_points(isolation_time, migration_time, coal_rate)
        return coalescence_points(np.concatenate([mig_bp, anc_bp]), coal_rate)

    def build_ctmc_system(self, isolation_time, migration_time, coal_rate, recomb_rate, mig_rate):
        iso, mig, single = self.isolation_state_space, self.migration_state_space, self.single_state_space
        Qi = iso.rate_matrix(isolation_rates(coal_rate, coal_rate, recomb_rate))
        Qm = mig.rate_matrix(migration_rates(coal_rate, coal_rate, recomb_rate, mig_rate, mig_rate))
        Qs = single.rate_matrix(single_rates(coal_rate, recomb_rate))
        mig_bp, anc_bp = self._break_points(isolation_time, migration_time, coal_rate)
        start = expm(Qi * mig_bp[0])[iso.i12_index] @ self._iso_to_mig
        pieces = _pieces_for([Qm] * len(mig_bp), mig_bp)
        pieces.append((Qm, anc_bp[0] - mig_bp[-1], self._mig_to_single))
        pieces += _pieces_for([Qs] * len(anc_bp), anc_bp)
        spaces = [mig] * self.no_mig_states + [single] * self.no_ancestral_states
        return PiecewiseCTMC(start, spaces, pieces)


class VariableCoalescenceRateIsolationModel(Model):
    """PSMC-like: one coalescence rate per epoch of intervals, optional split time
    (variable_coalescence_rate_isolation_model.py:83-171).

    Parameters: ``([split_time,] coal_rate per epoch ..., recomb_rate)``.
    """

    def __init__(self, intervals, est_split=False):
        self.isolation_state_space = isolation_space()
        self.single_state_space = single_space()
        self.intervals = intervals
        self.est_split = est_split
        self._iso_to_single = self.isolation_state_space.projection_to(self.single_state_space, True)

    def _unpack(self, parameters):
        if self.est_split:
            return parameters[0], parameters[1:-1], parameters[-1]
        return 0.0, parameters[0:-1], parameters[-1]

    def emission_points(self, *parameters):
        split_time, coal_rates, _ = self._unpack(parameters)
        break_points = psmc_break_points(sum(self.intervals), offset=split_time)
        return coalescence_points(break_points, _repeat_per_epoch(coal_rates, self.intervals))

    def build_ctmc_system(self, *parameters):
        split_time, coal_rates, recomb_rate = self._unpack(parameters)
        iso, single = self.isolation_state_space, self.single_state_space
        Qi = iso.rate_matrix(isolation_rates(coal_rates[0], coal_rates[0], recomb_rate))
        per_epoch = [single.rate_matrix(single_rates(c, recomb_rate)) for c in coal_rates]
        Qs = _repeat_per_epoch(per_epoch, self.intervals)
        bp = psmc_break_points(len(Qs), offset=split_time)
        start = expm(Qi * bp[0])[iso.i12_index] @ self._iso_to_single
        return PiecewiseCTMC(start, [single] * len(Qs), _pieces_for(Qs, bp))


class VariableCoalAndMigrationRateModel(Model):
    """Two populations with per-epoch coalescence and migration rates throughout
    (variable_migration_model.py:73-174).

    Parameters: ``(coal_1 per epoch, coal_2 per epoch, mig_12 per epoch, mig_21 per epoch, recomb)``.

    The reference hands its rates to ``make_rates_table_migration`` in the order
    (coal_1, coal_2, mig_12, mig_21, recomb) although that function is declared as
    (coal_1, coal_2, recomb, mig_12, mig_21) (variable_migration_model.py:163-165 against
    state_spaces.py:120-121), so the CTMC it actually builds recombines at ``mig_12``, migrates
    1->2 at ``mig_21`` and 2->1 at ``recomb``.  ``reference_rate_order=True`` (default) reproduces
    that, so results are identical to the reference's; ``False`` uses the documented meaning.
    """
    INITIAL_11 = 0
    INITIAL_12 = 1
    INITIAL_22 = 2

    def __init__(self, initial_configuration, intervals, reference_rate_order=True):
        self.migration_state_space = migration_space()
        sp = self.migration_state_space
        try:
            self.initial_state = {self.INITIAL_11: sp.i11_index, self.INITIAL_12: sp.i12_index,
                                  self.INITIAL_22: sp.i22_index}[initial_configuration]
        except KeyError:
            raise ValueError("initial_configuration must be INITIAL_11, INITIAL_12 or INITIAL_22")
        self.intervals = intervals
        self.no_states = sum(intervals)
        self.reference_rate_order = reference_rate_order

    def unpack_parameters(self, parameters):
        k = len(self.intervals)
        return (parameters[0:k], parameters[k:2 * k], parameters[2 * k:3 * k], parameters[3 * k:4 * k],
                parameters[-1])

    def emission_points(self, *parameters):
        coal_1, coal_2, _, _, _ = self.unpack_parameters(parameters)
        mean_rates = [(c1 + c2) / 2.0 for c1, c2 in zip(coal_1, coal_2)]
        return coalescence_points(psmc_break_points(self.no_states), _repeat_per_epoch(mean_rates, self.intervals))

    def build_ctmc_system(self, *parameters):
        coal_1, coal_2, mig_12, mig_21, recomb = self.unpack_parameters(parameters)
        sp = self.migration_state_space
        per_epoch = []
        for e in range(len(self.intervals)):
            if self.reference_rate_order:
                table = migration_rates(coal_1[e], coal_2[e], mig_12[e], mig_21[e], recomb)
            else:
                table = migration_rates(coal_1[e], coal_2[e], recomb, mig_12[e], mig_21[e])
            per_epoch.append(sp.rate_matrix(table))
        Qs = _repeat_per_epoch(per_epoch, self.intervals)
        bp = psmc_break_points(self.no_states)
        start = np.zeros(sp.size)
        start[self.initial_state] = 1.0          # upto_0 is the identity (variable_migration_model.py:63)
        return PiecewiseCTMC(start, [sp] * self.no_states, _pieces_for(Qs, bp))


class IsolationMigrationEpochsModel(Model):
    """Isolation-with-migration with ``no_epochs`` rate epochs in the migration and in the
    ancestral phase (isolation_with_migration_model_epochs.py:130-203).

    Parameters: ``(isolation_time, migration_time, recomb_rate, coal_rate x (2 no_epochs + 1),
    mig_rate x no_epochs)``.
    """

    def __init__(self, no_epochs, no_mig_states, no_ancestral_states):
        self.isolation_state_space = isolation_space()
        self.migration_state_space = migration_space()
        self.single_state_space = single_space()
        self.no_epochs = no_epochs
        self.no_mig_states = no_mig_states
        self.no_ancestral_states = no_ancestral_states
        self._iso_to_mig = self.isolation_state_space.projection_to(self.migration_state_space, False)
        self._mig_to_single = self.migration_state_space.projection_to(self.single_state_space, True)

    def _split(self, parameters):
        isolation_time, migration_time, recomb_rate = parameters[:3]
        coal_rates = parameters[3:2 * self.no_epochs + 1 + 3]
        mig_rates = parameters[2 * self.no_epochs + 1 + 3:]
        return isolation_time, migration_time, recomb_rate, coal_rates, mig_rates

    def emission_points(self, *parameters):
        isolation_time, migration_time, _, coal_rates, _ = self._split(parameters)
        tau1, tau2 = isolation_time, isolation_time + migration_time
        coal_rate = np.mean(coal_rates)                     # all epochs here (:156), unlike build_ctmc_system
        mig_bp = uniform_break_points(self.no_epochs * self.no_mig_states, tau1, tau2)
        anc_bp = exp_break_points(self.no_epochs * self.no_ancestral_states, coal_rate, tau2)
        return coalescence_points(np.concatenate([mig_bp, anc_bp]), coal_rate)

    def build_ctmc_system(self, *parameters):
        isolation_time, migration_time, recomb_rate, coal_rates, mig_rates = self._split(parameters)
        if len(coal_rates) != self.no_epochs * 2 + 1:
            raise ValueError("Isolation + #Epochs migration + #Epochs ancestral coalescence rates expected")
        if len(mig_rates) != self.no_epochs:
            raise ValueError("#Epochs migration rates expected")
        iso, mig, single = self.isolation_state_space, self.migration_state_space, self.single_state_space
        Qi = iso.rate_matrix(isolation_rates(coal_rates[0], coal_rates[0], recomb_rate))
        Qm = [mig.rate_matrix(migration_rates(coal_rates[e + 1], coal_rates[e + 1], recomb_rate,
                                              mig_rates[e], mig_rates[e])) for e in range(self.no_epochs)]
        Qa = [single.rate_matrix(single_rates(coal_rates[e + self.no_epochs + 1], recomb_rate))
              for e in range(self.no_epochs)]
        Qm = _repeat_per_epoch(Qm, [self.no_mig_states] * self.no_epochs)
        Qa = _repeat_per_epoch(Qa, [self.no_ancestral_states] * self.no_epochs)
        tau1, tau2 = isolation_time, isolation_time + migration_time
        mig_bp = uniform_break_points(self.no_epochs * self.no_mig_states, tau1, tau2)
        anc_bp = exp_break_points(self.no_epochs * self.no_ancestral_states,
                                  np.mean(coal_rates[self.no_epochs + 1:]), tau2)
        start = expm(Qi * mig_bp[0])[iso.i12_index] @ self._iso_to_mig
        pieces = _pieces_for(Qm, mig_bp)
        pieces.append((Qm[-1], anc_bp[0] - mig_bp[-1], self._mig_to_single))
        pieces += _pieces_for(Qa, anc_bp)
        spaces = [mig] * len(mig_bp) + [single] * len(anc_bp)
        return PiecewiseCTMC(start, spaces, pieces)
