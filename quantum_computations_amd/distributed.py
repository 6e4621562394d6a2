"""Registers sharded over the GPUs of one node: one process per GPU, ``torch.distributed`` (RCCL over xGMI).

The 2^n amplitudes are split by their top index bits: with G = 2^g ranks, rank r holds the contiguous shard of
amplitudes whose g most significant *physical* index bits spell r (reference qubits 0..g-1 at start, because
the reference is big-endian -- SURVEY.md 8e).  A gate is then one of:

* **local** -- every leg sits on a local bit: each rank runs the single-GPU kernel on its shard, no traffic;
* **block-diagonal in a remote leg** (diagonal gates, controls: Z, RZ, T, CZ, the control of CX, ...): the rank's
  own bit value selects the sub-block to apply locally -- still no traffic;
* **mixing a remote leg** (H, X, a Haar gate, the target of CX on a remote qubit): that qubit is first made
  local by an exchange step (grouped ``ncclSend``/``ncclRecv`` through ``batch_isend_irecv``), after which the gate
  is local.  With two ranks that is the pairwise swap of half a shard with rank ``r ^ 1``.  With four or more ranks
  ALL g rank bits are swapped with g local bits in one all-to-all: every rank keeps 1/G of its shard and sends 1/G
  to each of the other G - 1 ranks, so all of a GPU's xGMI links carry shard/G at the same time -- the time of ONE
  pairwise half-shard swap routed over all links, for g qubits instead of one (mitigation 1).  The
  qubits that leave are the local qubits whose next mixing use lies farthest ahead (``prepare``).  Swaps are *not*
  undone: the register keeps a logical -> physical bit map, so later gates on those qubits stay local.

Every exchange moves the data in pieces of at most ``chunk_amps`` amplitudes (1 GiB) through a two-piece staging
buffer: while piece c is on the links, piece c-1 is copied into place, so the staging memory stays at 2 GiB however
large the shard is (a 34-qubit register has 32 GiB shards).

**Gates ride inside the exchange** (round 3).  A gate whose mixing legs all sit on local bits below the slice size of
an exchange commutes with it slice by slice: ``run_circuit`` holds such gates back (they are local and would have run
just before the exchange anyway) and ``_exchange_bits`` applies them to every slice right after it has landed, on the
compute stream, while the next slice is on the links -- their HBM passes cost no wall time as long as they fit under the
transfer.  ``gates_in_exchanges`` counts them.

The class is written against a small "local engine" interface (the methods of ``DeviceState`` it uses) and
against torch tensors for the exchange, so the sharding logic runs unchanged on CPU tensors with the ``gloo``
backend in the tests (with a test-owned engine) and on HBM tensors with ``nccl`` in production.
"""
from __future__ import annotations

import numpy as np


def _default_engine_factory(device: int):
    import torch

    from .device import DeviceState

    def make(buf, n_local):
        stream = torch.cuda.current_stream(device).cuda_stream
        return DeviceState.view(n_local, buf.data_ptr(), buf.numel(), device=device, stream=stream, keepalive=buf)
    return make


_SWAP = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.complex128)


def _leg_is_block_diagonal(m: np.ndarray, k: int, leg: int) -> bool:
    """True if the 2^k x 2^k matrix never maps leg = a to leg = b != a (so the leg's bit value is conserved)."""
    t = m.reshape((2,) * (2 * k))
    idx01 = [slice(None)] * (2 * k)
    idx01[leg], idx01[k + leg] = 0, 1
    idx10 = list(idx01)
    idx10[leg], idx10[k + leg] = 1, 0
    return not np.any(t[tuple(idx01)]) and not np.any(t[tuple(idx10)])


def _restrict_leg(m: np.ndarray, k: int, leg: int, value: int) -> np.ndarray:
    """Sub-block of a matrix that is block-diagonal in ``leg``, for leg bit = ``value``: a 2^(k-1) matrix."""
    t = m.reshape((2,) * (2 * k))
    idx = [slice(None)] * (2 * k)
    idx[leg], idx[k + leg] = value, value
    return np.ascontiguousarray(t[tuple(idx)]).reshape(1 << (k - 1), 1 << (k - 1))


CHUNK_AMPS = 1 << 26      # staging piece: 2^26 amplitudes = 1 GiB (bounded exchange buffers)
RIDE_MIN_SLICE_AMPS = 1 << 20   # with riders an exchange is cut into at least four steps, of slices of >= 16 MiB
RIDE_MIN_BITS = 12        # no riding in slices below 2^12 amplitudes (64 KiB): such launches only cost host time
MAX_DEFERRED = 64         # gates held back for the next exchange at any time


class _DryEngine:
    """Stand-in local engine of a layout-only ``ShardedState`` (``ShardedState.plan_only``): accepts every call and
    does nothing, so the exchange schedule of a circuit can be computed without a register or a process group."""

    def __getattr__(self, name):
        def method(*args, **kwargs):
            if name == "measure_probs":
                return 0.5, 0.5
            if name in ("norm2", "fill_random"):
                return 1.0
            return self
        return method


class ShardedState:
    """n-qubit complex128 register sharded over ``world`` ranks (a power of two)."""

    ndim = 1

    def __init__(self, n_qubits: int, buf, engine_factory, group=None, *, chunk_amps: int | None = None,
                 policy: str = "auto", _dry: tuple[int, int] | None = None):
        """``policy``: which exchange brings a remote qubit into the shards -- ``"auto"`` (all rank bits at once when
        there are four or more ranks, the pairwise half-shard swap otherwise), ``"pairwise"`` (always one qubit)."""
        if _dry is None:
            import torch.distributed as dist

            self._dist = dist
            self.group = group
            self.world = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
        else:
            self._dist, self.group = None, None
            self.world, self.rank = _dry
        self.g = (self.world - 1).bit_length()
        if 1 << self.g != self.world:
            raise ValueError("the register shards over a power-of-two number of ranks")
        if n_qubits < self.g + 1:
            raise ValueError("need at least one local qubit per rank")
        if policy not in ("auto", "pairwise"):
            raise ValueError("policy must be 'auto' or 'pairwise'")
        self.n = n_qubits
        self.n_local = n_qubits - self.g
        self.buf = buf                                    # torch tensor, 2^n_local complex128, this rank's shard
        self._factory = engine_factory
        self.local = _DryEngine() if _dry is not None else engine_factory(buf, self.n_local)
        self._scratch = None                              # two staging pieces, allocated on first exchange
        import os
        self.chunk_amps = int(chunk_amps or os.environ.get("QSV_EXCHANGE_CHUNK_AMPS", CHUNK_AMPS))
        if self.chunk_amps < 1 or self.chunk_amps & (self.chunk_amps - 1):
            raise ValueError("chunk_amps must be a power of two")
        self.policy = policy
        # physical bit position of each logical bit (logical bit b = reference qubit n-1-b)
        self.phys = list(range(n_qubits))
        self.exchanges = 0                                # exchange steps performed (for reports)
        self.qubits_exchanged = 0                         # rank bits swapped in by those steps
        self.bytes_sent = 0                               # bytes this rank put on the links
        self.link_bytes = 0                               # bytes over its busiest link (xGMI is point to point: this
                                                          # is what an exchange's duration is proportional to)
        self.messages = 0                                 # point-to-point messages this rank sent
        self.local_swaps = 0                              # local passes spent lining victims up for an exchange
        self.overlap = os.environ.get("QSV_EXCHANGE_OVERLAP", "1") != "0"
        self._host_sync = os.environ.get("QSV_EXCHANGE_HOST_SYNC", "0") == "1"   # debugging aid: drain before sending
        self.ride_min_slice = int(os.environ.get("QSV_RIDE_MIN_SLICE_AMPS", RIDE_MIN_SLICE_AMPS))
        self.ride_min_bits = int(os.environ.get("QSV_RIDE_MIN_BITS", RIDE_MIN_BITS))
        self.last_exchange = None                         # description of the most recent exchange slice issued
        self.gates_in_exchanges = 0                       # gates applied slice by slice inside exchange steps
        self.rider_launches = 0                           # per-slice launches those gates took
        self._deferred = []                               # gates held back by run_circuit to ride in the next exchange
        self._apply_cb = None                             # how run_circuit's caller applies a gate (flushes use it)
        self._advanced_already = False                    # set while held-back gates run: the plan knows them already
        self._window_engine = None                        # engine re-pointed at one slice after the other
        self._plan = None                                 # look-ahead set by prepare(): [(indices, mixing qubits)]
        self._cursor = 0                                  # first plan entry not applied yet
        self._done = []                                   # plan entries already applied (they may arrive out of order)

    # ---- construction ---------------------------------------------------------------------------
    @classmethod
    def plan_only(cls, n_qubits: int, world: int, policy: str = "auto") -> "ShardedState":
        """A register without data: gates only update the qubit layout and the exchange counters.  The layout logic
        never depends on the rank, so this tells what every rank of a ``world``-rank run will do (bench reports,
        tests of the scheduling policy)."""
        return cls(n_qubits, None, None, policy=policy, _dry=(world, 0))

    @classmethod
    def random(cls, n_qubits: int, seed: int, device: int = 0, group=None) -> "ShardedState":
        """Normalised pseudo-random register generated shard by shard on the GPUs (counter-based)."""
        import torch
        import torch.distributed as dist

        world = dist.get_world_size(group)
        n_local = n_qubits - (world - 1).bit_length()
        buf = torch.empty(1 << n_local, dtype=torch.complex128, device=torch.device("cuda", device))
        st = cls(n_qubits, buf, _default_engine_factory(device), group)
        st.fill_random(seed)
        return st

    @classmethod
    def zeros(cls, n_qubits: int, device: int = 0, group=None) -> "ShardedState":
        import torch
        import torch.distributed as dist

        world = dist.get_world_size(group)
        n_local = n_qubits - (world - 1).bit_length()
        buf = torch.zeros(1 << n_local, dtype=torch.complex128, device=torch.device("cuda", device))
        st = cls(n_qubits, buf, _default_engine_factory(device), group)
        if st.rank == 0:
            st.local.set_basis(0)
        return st

    def set_basis(self, index: int) -> None:
        """The basis state ``|index>`` (reference bit order); the layout goes back to the identity map."""
        if not 0 <= index < (1 << self.n):
            raise ValueError("basis index out of range")
        self.phys = list(range(self.n))
        if index >> self.n_local == self.rank:
            self.local.set_basis(index & ((1 << self.n_local) - 1))
        else:
            self.local.apply_scale(0.0)

    def warm_up_links(self) -> None:
        """One tiny message to and from every other rank: RCCL sets up its per-peer channels on first use, which
        would otherwise be charged to the first exchange of a circuit."""
        if self._dist is None or self.world == 1:
            return
        import torch

        out = torch.zeros(self.world, dtype=self.buf.dtype, device=self.buf.device)
        inc = torch.empty_like(out)
        others = [r for r in range(self.world) if r != self.rank]
        self._p2p([(r, out[r:r + 1]) for r in others], [(r, inc[r:r + 1]) for r in others]).wait()
        self._allreduce_sum([0.0])

    def fill_random(self, seed: int) -> None:
        n2 = self.local.fill_random(seed, index_offset=self.rank << self.n_local, normalise=False)
        self.local.apply_scale(1.0 / float(np.sqrt(self._allreduce_sum([n2])[0])))
        self.phys = list(range(self.n))

    # ---- bookkeeping ----------------------------------------------------------------------------
    @property
    def num_qubits(self) -> int:
        return self.n

    @property
    def num_amps(self) -> int:
        return 1 << self.n

    @property
    def shape(self):
        return (self.num_amps,)

    def _bit(self, qubit: int) -> int:
        if not 0 <= qubit < self.n:
            raise ValueError(f"qubit index {qubit} out of range for a {self.n}-qubit register")
        return self.phys[self.n - 1 - qubit]

    def _rank_bit(self, phys_bit: int) -> int:
        return (self.rank >> (phys_bit - self.n_local)) & 1

    def _local_qubit(self, phys_bit: int) -> int:
        return self.n_local - 1 - phys_bit

    def remote_qubits(self) -> list[int]:
        """Reference indices of the qubits that currently sit on rank bits."""
        return sorted(self.n - 1 - lb for lb, p in enumerate(self.phys) if p >= self.n_local)

    def sync(self) -> None:
        self.local.sync()

    # ---- the exchange step ----------------------------------------------------------------------
    def _swap_bits_local(self, bit_a: int, bit_b: int) -> None:
        if bit_a == bit_b:
            return
        self.local.apply_swap(self._local_qubit(bit_a), self._local_qubit(bit_b))
        self.local_swaps += 1
        ia, ib = self.phys.index(bit_a), self.phys.index(bit_b)
        self.phys[ia], self.phys[ib] = bit_b, bit_a

    # The three collectives the register needs.  Kept as small methods so that a test can stage them through
    # host memory (gloo on a box with fewer GPUs than ranks); production uses them as written, on RCCL.
    def _p2p(self, sends, recvs):
        """Start one grouped point-to-point step: ``sends`` / ``recvs`` are lists of ``(peer rank, tensor)``.
        Returns an object with ``wait()``.  On RCCL this is one ncclGroupStart/End of ncclSend/ncclRecv on the
        communicator's own stream; ``wait()`` makes the current stream wait for it (no host block), so the caller's
        next copy kernel overlaps with whatever group is issued after this one."""
        import torch

        dist = self._dist
        ops = []
        for peer, t in sends:
            p = dist.get_global_rank(self.group, peer) if self.group is not None else peer
            # RCCL has no complex dtype: move the amplitudes as (re, im) float64 pairs (same bytes, no copy)
            ops.append(dist.P2POp(dist.isend, torch.view_as_real(t), p, self.group))
        for peer, t in recvs:
            p = dist.get_global_rank(self.group, peer) if self.group is not None else peer
            ops.append(dist.P2POp(dist.irecv, torch.view_as_real(t), p, self.group))
        works = dist.batch_isend_irecv(ops)

        class _Step:
            @staticmethod
            def wait():
                for w in works:
                    w.wait()
        return _Step

    def _allreduce_sum(self, values: list[float]) -> list[float]:
        if self._dist is None:
            return [float(v) for v in values]
        import torch

        t = torch.tensor(values, dtype=torch.float64, device=self.buf.device)
        self._dist.all_reduce(t, group=self.group)
        return [float(v) for v in t.cpu()]

    def _broadcast(self, tensor, src: int) -> None:
        """Rank ``src``'s ``tensor`` (complex128, this register's device) to every rank, in place."""
        import torch

        dist = self._dist
        root = dist.get_global_rank(self.group, src) if self.group is not None else src
        dist.broadcast(torch.view_as_real(tensor), root, group=self.group)

    def _slice_amps(self, k: int, riding: bool) -> int:
        """Amplitudes per slice of one piece in an exchange of ``k`` rank bits: the staging piece shared by the
        2^k - 1 peers, rounded down to a power of two; with gates riding along at least four steps (something to
        overlap with) of slices no smaller than ``ride_min_slice``."""
        parts = 1 << k
        piece = 1 << (self.n_local - k)
        cs = max(1, self.chunk_amps // (parts - 1))
        cs = min(1 << (cs.bit_length() - 1), piece)           # power of two, so it divides the piece
        if riding:
            cs = min(cs, max(piece >> 2, min(piece, self.ride_min_slice)))
        return cs

    def _rider_bits(self) -> int:
        """Local bits below this position stay inside one slice of any exchange this register may run."""
        k = self.g if self.policy == "auto" and self.g >= 2 else 1
        if self.n_local - k < 1:
            return 0
        return self._slice_amps(k, True).bit_length() - 1

    def _exchange_bits(self, gbits: list[int]) -> None:
        """Swap the rank bits ``gbits`` (ascending physical positions) with the top ``len(gbits)`` local bits.

        The 2^k ranks that differ only in ``gbits`` form a subgroup; the shard is 2^k contiguous pieces (one per
        value j of its top k local bits).  Piece j goes to the subgroup member whose ``gbits`` spell j, and that
        member's piece number <my gbits value> comes back into the same slot; the piece with j = my own value stays.
        k = 1 is the pairwise half-shard swap, k = g the all-to-all over every link.  The pieces travel in slices of
        at most ``chunk_amps / (2^k - 1)`` amplitudes through two staging slices: slice c is on the links while
        slice c-1 is copied into place -- and while the gates held back by ``run_circuit`` (``_deferred``) are applied
        to the slices that have landed."""
        k = len(gbits)
        parts = 1 << k
        piece = 1 << (self.n_local - k)
        mine = sum(self._rank_bit(gb) << i for i, gb in enumerate(gbits))
        mask = sum(1 << (gb - self.n_local) for gb in gbits)
        peers = [(self.rank & ~mask) | sum(((j >> i) & 1) << (gb - self.n_local) for i, gb in enumerate(gbits))
                 for j in range(parts)]
        others = [j for j in range(parts) if j != mine]
        riders, self._deferred = self._deferred, []
        cs = self._slice_amps(k, bool(riders))
        steps = piece // cs
        self.exchanges += 1
        self.qubits_exchanged += k
        self.bytes_sent += len(others) * piece * 16
        self.link_bytes += piece * 16
        self.messages += len(others) * steps
        # the logical bit that lived on top-local slot i now lives on gbits[i], and vice versa (the riders below are
        # resolved against the layout the landed data has)
        for i, gb in enumerate(gbits):
            slot = self.n_local - k + i
            i_slot, i_g = self.phys.index(slot), self.phys.index(gb)
            self.phys[i_slot], self.phys[i_g] = gb, slot
        ride = self._prepare_riders(riders, cs) if riders else None
        if self.buf is None:
            return
        import torch

        stage = len(others) * cs
        if self._scratch is None or self._scratch.numel() < 2 * stage:
            self._scratch = torch.empty(2 * stage, dtype=self.buf.dtype, device=self.buf.device)
        if self._host_sync:
            self.local.sync()      # not needed for ordering: the collectives queue behind the current stream

        def land(step, c, half):
            step.wait()
            for slot, j in enumerate(others):
                self.buf[j * piece + c * cs:j * piece + (c + 1) * cs].copy_(half[slot * cs:(slot + 1) * cs])
            if ride is not None:
                for j in others:
                    ride(j * piece + c * cs)

        pending = None
        for c in range(steps):
            half = self._scratch[(c & 1) * stage:(c & 1) * stage + stage]
            sends = [(peers[j], self.buf[j * piece + c * cs:j * piece + (c + 1) * cs]) for j in others]
            recvs = [(peers[j], half[slot * cs:(slot + 1) * cs]) for slot, j in enumerate(others)]
            # what a failure report names (bench.py fail_rank): the step a stalled collective belongs to
            self.last_exchange = {"step": self.exchanges, "rank_bits": list(gbits), "slice": f"{c + 1} of {steps}",
                                  "slice_MiB": cs * 16 / 2**20, "peers": [peers[j] for j in others]}
            step = self._p2p(sends, recvs)             # slice c is on the links ...
            if pending is not None:
                land(*pending)                         # ... while slice c-1 is copied into place and takes its gates
            if ride is not None:
                ride(mine * piece + c * cs)            # the piece that stays: its slices take theirs along the way
            pending = (step, c, half)
        land(*pending)

    # ---- gates that ride inside an exchange ----------------------------------------------------------
    def _window(self, view, n_window: int):
        """A local engine on ``view`` (a contiguous window of the shard holding 2^n_window amplitudes)."""
        w = self._window_engine
        if w is not None:
            w.rebind(n_window, view.data_ptr(), view.numel(), keepalive=view)
            return w
        w = self._factory(view, n_window)
        if hasattr(w, "rebind"):
            self._window_engine = w                    # one handle walks the slices (qsv_rebind_view)
        return w

    def _prepare_riders(self, riders, cs: int):
        """Returns ``ride(offset)``: apply every rider to the slice of ``cs`` amplitudes at ``offset`` of the shard.

        A rider's legs are looked up in the layout the exchange leaves behind.  Legs on bits inside the slice are
        real legs of the launch; the others are conserved by the gate (that is what made it a rider) and their bit
        values are fixed over the slice -- by this rank's id (rank bits) or by the slice's offset (local bits above
        the slice) -- so they only select a sub-block of the matrix."""
        nw = cs.bit_length() - 1
        prepared = []
        for gate in riders:
            qubits = [int(q) for q in gate.indices]
            m = np.asarray(gate.matrix, dtype=np.complex128)
            bits = [self._bit(q) for q in qubits]
            if any(b >= nw for q, b in zip(qubits, bits) if q in self.mixing_qubits(gate)):
                raise AssertionError("a deferred gate lost its place inside the slices")   # _settle_deferred guards this
            prepared.append((m, bits, {}))
        self.gates_in_exchanges += len(prepared)

        def ride(offset: int) -> None:
            eng = self._window(self.buf[offset:offset + cs], nw)
            for m, bits, cache in prepared:
                fixed = tuple(self._rank_bit(b) if b >= self.n_local else (offset >> b) & 1 if b >= nw else -1
                              for b in bits)
                hit = cache.get(fixed)
                if hit is None:
                    sub, legs = m, list(bits)
                    j = 0
                    for value in fixed:
                        if value < 0:
                            j += 1
                        else:
                            sub = _restrict_leg(sub, len(legs), j, value)
                            del legs[j]
                    if not legs:
                        hit = ("scale", complex(sub[0, 0])) if sub[0, 0] != 1.0 else ("skip", None)
                    elif np.array_equal(sub, np.identity(sub.shape[0])):
                        hit = ("skip", None)
                    else:
                        hit = ("matrix", (sub, [nw - 1 - b for b in legs]))
                    cache[fixed] = hit
                kind, what = hit
                if kind == "matrix":
                    eng.apply_matrix(*what)
                    self.rider_launches += 1
                elif kind == "scale":
                    eng.apply_scale(what)
                    self.rider_launches += 1
        return ride

    def _can_ride(self, gate) -> bool:
        """May ``gate`` (local in the current layout) be held back for the next exchange?  Every leg it mixes must
        sit below the slice size; conserved legs may sit anywhere."""
        if not self.overlap or self.world == 1 or len(self._deferred) >= MAX_DEFERRED:
            return False
        matrix = getattr(gate, "matrix", None)
        mixing = self.mixing_qubits(gate)
        if matrix is None or mixing is None or len(gate.indices) > 5:
            return False
        if len(gate.indices) == 2 and np.array_equal(np.asarray(matrix), _SWAP):
            return False                                                 # a relabelling: nothing to hide
        rb = self._rider_bits()
        return rb >= self.ride_min_bits and all(self._bit(q) < rb for q in mixing)

    def _apply_now(self, gate) -> None:
        if self._apply_cb is not None:
            self._apply_cb(gate)
        else:
            gate.apply(self)

    def _flush_deferred(self, count: int | None = None, nested: bool = False) -> None:
        """Apply the first ``count`` held-back gates (default: all) the ordinary way, in order.  ``nested``: called
        from inside another gate's application -- the caller's per-gate callback must not be re-entered."""
        held = self._deferred if count is None else self._deferred[:count]
        self._deferred = [] if count is None else self._deferred[count:]
        self._advanced_already = True          # _dispatch told the look-ahead when it held the gate back
        try:
            for gate in held:
                if nested:
                    gate.apply(self)
                else:
                    self._apply_now(gate)
        finally:
            self._advanced_already = False

    def _settle_deferred(self, victims: list[int], k: int) -> None:
        """Before an exchange of ``k`` rank bits that gives up the local bits ``victims``: held-back gates that mix
        a qubit which is about to leave, or which no longer fits below this exchange's slice size, run now --
        together with everything held back before them (program order among overlapping gates)."""
        if not self._deferred:
            return
        nw = self._slice_amps(k, True).bit_length() - 1
        last = -1
        for i, gate in enumerate(self._deferred):
            for q in self.mixing_qubits(gate):
                b = self._bit(q)
                if b in victims or b >= nw:
                    last = i
        if last >= 0:
            self._flush_deferred(last + 1, nested=True)

    # ---- look-ahead: which local qubits to give up --------------------------------------------------
    @staticmethod
    def mixing_qubits(gate) -> frozenset | None:
        """The qubits ``gate`` does not conserve (those must be local when it runs); ``None`` for gates that change
        the register size or have no square matrix.  Objects may state theirs in a ``mixing`` attribute (a
        multi-controlled phase mixes nothing); the answer is cached on the gate."""
        inner = getattr(gate, "gate", gate)                         # ClassicalControl wraps a gate
        cached = getattr(inner, "_qsv_mixing", None)
        if cached is not None:
            return cached
        explicit = getattr(inner, "mixing", None)
        if explicit is not None:
            mixing = frozenset(explicit)
        else:
            matrix, indices = getattr(inner, "matrix", None), list(getattr(inner, "indices", []))
            if matrix is None or np.asarray(matrix).shape[0] != np.asarray(matrix).shape[1]:
                return None
            m = np.asarray(matrix, dtype=np.complex128)
            k = len(indices)
            if k == 2 and np.array_equal(m, _SWAP):
                mixing = frozenset()                                 # a relabelling of the qubit map
            else:
                mixing = frozenset(q for j, q in enumerate(indices) if not _leg_is_block_diagonal(m, k, j))
        try:
            inner._qsv_mixing = mixing
        except AttributeError:
            pass
        return mixing

    def prepare(self, circuit) -> None:
        """Tell the register which gates are about to be applied (objects with ``indices`` / ``matrix``, in order).

        With the plan, an exchange evicts the local qubits whose next *mixing* use lies farthest ahead (Belady's
        rule) instead of whichever qubits sit on the top local bits, which cuts the number of exchanges of a random
        circuit several-fold.  Purely an optimisation: every rank computes the same plan from the same
        circuit, gates may arrive in a different order as long as they are the announced ones (``run_circuit`` applies
        commuting gates local-first), a gate that is not in the plan simply switches the look-ahead off, and the plan
        stops at the first measurement / insertion (they renumber the qubits)."""
        plan = []
        for gate in circuit:
            mixing = self.mixing_qubits(gate)
            if mixing is None:
                break
            inner = getattr(gate, "gate", gate)
            indices = tuple(int(q) for q in inner.indices)
            matrix = getattr(inner, "matrix", None)
            relabel = len(indices) == 2 and matrix is not None and np.array_equal(np.asarray(matrix), _SWAP)
            plan.append((indices, mixing, relabel))
        self._plan, self._cursor, self._done = plan, 0, [False] * len(plan)

    PLAN_WINDOW = 512      # how far ahead of the first pending entry an announced gate may arrive
    PLAN_HORIZON = 4096    # how far ahead the eviction rule looks for a qubit's next use (beyond: "far")

    def _advance_plan(self, indices) -> None:
        if self._plan is None or self._advanced_already:
            return
        want = tuple(int(q) for q in indices)
        for step in range(self._cursor, min(len(self._plan), self._cursor + self.PLAN_WINDOW)):
            if not self._done[step] and self._plan[step][0] == want:
                self._done[step] = True
                while self._cursor < len(self._plan) and self._done[self._cursor]:
                    self._cursor += 1
                return
        self._plan = None                                            # the caller left the announced circuit

    def _next_mixing_use(self, qubit: int) -> int:
        """First pending plan step that mixes the data now known as ``qubit`` (SWAPs ahead rename it on the way)."""
        for step in range(self._cursor, min(len(self._plan), self._cursor + self.PLAN_HORIZON)):
            if self._done[step]:
                continue
            indices, mixing, relabel = self._plan[step]
            if relabel:
                if qubit == indices[0]:
                    qubit = indices[1]
                elif qubit == indices[1]:
                    qubit = indices[0]
            elif qubit in mixing:
                return step
        return 1 << 60

    def needs_exchange(self, gate) -> bool:
        """Would applying ``gate`` now move data between GPUs?  (Same answer on every rank: layout only.)"""
        mixing = self.mixing_qubits(gate)
        if mixing is None:
            inner = getattr(gate, "gate", gate)
            mixing = frozenset(getattr(inner, "indices", []))        # measurement / unknown: its qubits must be local
            if getattr(inner, "matrix", None) is not None and np.asarray(inner.matrix).shape[0] == 1:
                return False                                         # Insert: never any traffic
        return any(0 <= q < self.n and self._bit(q) >= self.n_local for q in mixing)

    def run_circuit(self, circuit, apply=None) -> list:
        """Apply ``circuit`` with commuting gates reordered local-first: of the leading gates that act on pairwise
        disjoint qubits (they commute), one that needs no exchange goes first; only when every one of them mixes a
        remote qubit does the first take its exchange -- by then the qubits that leave have had their gates of this
        layer.  A layer of one-qubit gates on all n qubits (Grover's H and X walls) then costs one exchange instead
        of one per remote qubit per wall.  Returns the gates in the order applied; ``apply(gate)`` does the
        application (default ``gate.apply(self)``) and may return False to say the gate was skipped."""
        from collections import deque
        from itertools import islice

        pending = deque(circuit)
        self.prepare(pending)
        order = []
        self._apply_cb = apply
        try:
            while pending:
                pick, used = 0, set()
                for j, gate in enumerate(islice(pending, 256)):
                    inner = getattr(gate, "gate", gate)
                    if inner is not gate or self.mixing_qubits(gate) is None:
                        break                                            # barriers are taken only from the front
                    qs = set(inner.indices)
                    if qs & used:
                        break
                    used |= qs
                    if not self.needs_exchange(gate):
                        pick = j
                        break
                pending.rotate(-pick)                                    # O(pick), whatever the length of the circuit
                gate = pending.popleft()
                pending.rotate(pick)
                order.append(gate)
                self._dispatch(gate)
            self._flush_deferred()
        finally:
            self._apply_cb = None
        return order

    def _dispatch(self, gate) -> None:
        """Apply the gate ``run_circuit`` picked -- or hold it back: a local gate whose mixing legs all lie inside
        the slices of an exchange runs just as well *inside* the next exchange step, slice by slice while the other
        slices travel (``_exchange_bits``).  Held-back gates keep their order; whatever would not commute with them
        (a gate on one of their qubits, a barrier) makes them run first."""
        inner = getattr(gate, "gate", gate)
        plain = inner is gate and self.mixing_qubits(gate) is not None
        if plain and not self.needs_exchange(gate) and self._can_ride(gate):
            # for the look-ahead the gate counts as applied: which qubits an exchange gives up -- the whole schedule
            # -- is the same with and without riders (a rider whose qubit is chosen to leave simply runs first)
            self._advance_plan(gate.indices)
            self._deferred.append(gate)
            return
        if self._deferred:
            if not plain:
                self._flush_deferred()                                   # measurement, insertion, classical control
            elif not self.needs_exchange(gate):
                held = set()
                for g in self._deferred:
                    held.update(g.indices)
                if held & set(gate.indices):
                    self._flush_deferred()
            # a gate that needs an exchange takes the held-back gates along: they run inside its exchange step,
            # before its own kernel
        self._apply_now(gate)

    def _choose_victims(self, count: int, avoid: set[int]) -> list[int]:
        """``count`` local bits whose qubits leave the shard, never one of ``avoid``."""
        free = [b for b in range(self.n_local) if b not in avoid]
        if len(free) < count:
            raise ValueError("gate has more legs than a shard has qubits")
        if self._plan is None:
            return sorted(free)[-count:]                             # the top bits need no local swap
        # farthest next use first; among equals the highest bit
        free.sort(key=lambda b: (self._next_mixing_use(self.n - 1 - self.phys.index(b)), b), reverse=True)
        return free[:count]

    def _localise(self, gbit: int, avoid: set[int]) -> int:
        """Bring the qubit on rank bit ``gbit`` into the shard (the local bits in ``avoid`` stay); returns the local
        bit it now occupies.  Every rank takes the same decisions: they depend on the layout and the plan only."""
        logical = self.phys.index(gbit)
        free = self.n_local - len(avoid)
        if self.policy == "auto" and self.g >= 2 and free >= self.g:
            gbits = list(range(self.n_local, self.n))                # all rank bits in one all-to-all
        else:
            gbits = [gbit]                                           # pairwise half-shard swap with rank r ^ bit
        k = len(gbits)
        victims = self._choose_victims(k, avoid)
        self._settle_deferred(victims, k)
        # the pieces are contiguous only if the victims sit on the top k local bits: line them up (local passes;
        # an exchange costs 30x more).  Victims already inside the top block keep their slot.
        top = list(range(self.n_local - k, self.n_local))
        outside = [v for v in victims if v not in top]
        for slot in top:
            if slot not in victims:
                self._swap_bits_local(slot, outside.pop())
        self._exchange_bits(gbits)
        return self.phys[logical]

    # ---- gates ----------------------------------------------------------------------------------
    #
    # Every decision that changes the data layout (which qubits get localised) depends only on the gate and
    # on the logical -> physical map, never on this rank's own bits: all ranks take the same path through
    # _localise, so the map stays identical everywhere and every exchange finds its partner.  Only the
    # *local* work after that depends on the rank (which sub-block of a conserved remote leg applies).
    def _localise_all(self, qubits: list[int], keep_local: list[int]) -> None:
        """Make every qubit of ``qubits`` local; ``keep_local`` are other legs that must not be displaced."""
        for q in qubits:
            b = self._bit(q)
            if b >= self.n_local:
                avoid = {self._bit(x) for x in list(qubits) + list(keep_local) if self._bit(x) < self.n_local}
                self._localise(b, avoid)

    def apply_matrix(self, matrix, indices) -> "ShardedState":
        """``U_full @ ket`` for a 2^k x 2^k matrix on reference qubits ``indices`` (``Gate.apply``)."""
        qubits = [int(q) for q in indices]
        k = len(qubits)
        m = np.asarray(matrix, dtype=np.complex128)
        if m.shape != (1 << k, 1 << k):
            raise ValueError("Dimensions of given matrix is not compatible with number of indices.")
        if len(set(qubits)) != k:
            raise ValueError("Indices must be distinct.")
        for q in qubits:
            self._bit(q)
        if k == 2 and np.array_equal(m, _SWAP):
            return self.apply_swap(*qubits)           # a relabelling of the qubit map: no data moves
        self._advance_plan(qubits)
        bits = [self._bit(q) for q in qubits]
        if all(b < self.n_local for b in bits):       # every leg inside the shard: nothing to decide
            self.local.apply_matrix(m, [self._local_qubit(b) for b in bits])
            return self
        conserved = [_leg_is_block_diagonal(m, k, j) for j in range(k)]
        mixing = [q for q, c in zip(qubits, conserved) if not c]
        if len(mixing) > self.n_local:
            raise ValueError("gate has more mixing legs than a shard has qubits")
        # 1) remote legs the gate mixes: bring those qubits into the shard (one half-shard exchange each)
        self._localise_all(mixing, [q for q in qubits if q not in mixing])
        # 2) remote legs whose bit value the gate conserves (diagonal legs, controls): this rank's sub-block
        j = 0
        while j < len(qubits):
            b = self._bit(qubits[j])
            if b >= self.n_local:
                m = _restrict_leg(m, len(qubits), j, self._rank_bit(b))
                del qubits[j]
            else:
                j += 1
        if not qubits:
            if m[0, 0] != 1.0:
                self.local.apply_scale(complex(m[0, 0]))
            return self
        if np.array_equal(m, np.identity(m.shape[0])):
            return self
        self.local.apply_matrix(m, [self._local_qubit(self._bit(q)) for q in qubits])
        return self

    def apply_diagonal(self, diagonal, indices) -> "ShardedState":
        return self.apply_matrix(np.diag(np.asarray(diagonal, dtype=np.complex128)), indices)

    def apply_cx(self, control: int, target: int) -> "ShardedState":
        return self.apply_controlled(np.array([[0, 1], [1, 0]], dtype=complex), [control], target)

    def apply_swap(self, q0: int, q1: int) -> "ShardedState":
        """SWAP is a relabelling of the logical -> physical map: no amplitude moves."""
        self._bit(q0), self._bit(q1)
        if q0 == q1:
            raise ValueError("Indices must be distinct.")
        self._advance_plan([q0, q1])
        b0, b1 = self.n - 1 - q0, self.n - 1 - q1
        self.phys[b0], self.phys[b1] = self.phys[b1], self.phys[b0]
        return self

    def apply_mcphase(self, qubits, phase: complex) -> "ShardedState":
        """Multiply the amplitudes whose ``qubits`` are all 1 by ``phase``: never any traffic."""
        bits = [self._bit(int(q)) for q in qubits]
        self._advance_plan([int(q) for q in qubits])
        if any(self._rank_bit(b) == 0 for b in bits if b >= self.n_local):
            return self
        local = [self._local_qubit(b) for b in bits if b < self.n_local]
        if local:
            self.local.apply_mcphase(local, phase)
        else:
            self.local.apply_scale(complex(phase))
        return self

    def apply_controlled(self, matrix, controls, target: int) -> "ShardedState":
        """2x2 ``matrix`` on ``target`` where all ``controls`` are 1.  Controls never move data; the target is
        localised (by every rank, see above) only if the matrix mixes it."""
        controls = [int(c) for c in controls]
        target = int(target)
        if len(set(controls + [target])) != len(controls) + 1:
            raise ValueError("Indices must be distinct.")
        u = np.asarray(matrix, dtype=np.complex128).reshape(2, 2)
        diagonal = u[0, 1] == 0 and u[1, 0] == 0
        for q in controls + [target]:
            self._bit(q)
        self._advance_plan(controls + [target])
        if not diagonal:
            self._localise_all([target], controls)
        cbits = [self._bit(c) for c in controls]
        if any(self._rank_bit(b) == 0 for b in cbits if b >= self.n_local):
            return self
        local_c = [self._local_qubit(b) for b in cbits if b < self.n_local]
        tb = self._bit(target)
        if tb >= self.n_local:                       # diagonal on a remote target: a phase picked by our bit
            phase = u[self._rank_bit(tb), self._rank_bit(tb)]
            if phase != 1.0:
                if local_c:
                    self.local.apply_mcphase(local_c, phase)
                else:
                    self.local.apply_scale(complex(phase))
        elif local_c:
            self.local.apply_controlled(u, local_c, self._local_qubit(tb))
        else:
            self.local.apply_matrix(u, [self._local_qubit(tb)])
        return self

    def insert(self, index: int, amplitudes) -> "ShardedState":
        """``Insert.apply`` (gates.py:145-153) on the sharded register: the new qubit (a product factor
        ``amplitudes = (a0, a1)``) becomes reference qubit ``index``.  It is made the *least significant local* qubit,
        so no amplitude crosses a link: every shard doubles in place (a new buffer of twice the size, the local engine
        interleaves the two scaled copies) and only the logical->physical map records where the qubit belongs."""
        import torch

        if index < 0 or index > self.n:
            raise ValueError("new_ordering must be a permutation of all qubits")
        self.local.sync()
        old_amps = 1 << self.n_local
        grown = torch.empty(2 * old_amps, dtype=self.buf.dtype, device=self.buf.device)
        grown[:old_amps].copy_(self.buf[:old_amps])
        self.buf = grown
        self.local = self._factory(grown, self.n_local)       # same n_local qubits, room for one more
        self.local.insert(self.n_local, amplitudes)            # local reference position n_local = physical bit 0
        new_logical_bit = self.n - index                       # bit of the new qubit in the (n+1)-qubit register
        self.phys = [p + 1 for p in self.phys]
        self.phys.insert(new_logical_bit, 0)
        self.n += 1
        self.n_local += 1
        self._plan = None
        return self

    # ---- measurement ----------------------------------------------------------------------------
    def measure_probs(self, index: int, eig0, eig1) -> tuple[float, float]:
        import torch

        b = self._bit(index)
        if b >= self.n_local:
            b = self._localise(b, set())
        p0, p1 = self._allreduce_sum(list(self.local.measure_probs(self._local_qubit(b), eig0, eig1)))
        return p0, p1

    def agree_on_outcome(self, outcome: int) -> int:
        """The measurement outcome every rank must use: rank 0's draw.

        ``M.apply`` draws from the process-global ``np.random`` like the reference (gates.py:183).  One process per
        GPU means one generator per rank, and nothing keeps their states equal; shards that collapsed onto different
        outcomes would be a silently wrong register (and diverging ``Simulator.results`` would send later exchanges
        to partners that never post them).  Every rank still draws -- ranks seeded alike stay in step with a
        single-process run of the same script -- but rank 0's bit is the one applied."""
        return int(round(self._allreduce_sum([float(outcome) if self.rank == 0 else 0.0])[0]))

    def collapse(self, index: int, eig, scale: float) -> "ShardedState":
        """Project reference qubit ``index`` on ``eig`` and drop it: every shard shrinks by half."""
        b = self._bit(index)
        if b >= self.n_local:
            b = self._localise(b, set())
        self.local.collapse(self._local_qubit(b), eig, scale)
        # physical bits above b move down by one; the measured logical bit disappears
        lb = self.phys.index(b)
        del self.phys[lb]
        self.phys = [p - 1 if p > b else p for p in self.phys]
        self.n -= 1
        self.n_local -= 1
        self._plan = None                       # the qubits are renumbered: any look-ahead is stale
        self.buf = self.buf[: 1 << self.n_local]
        return self

    # ---- read-out -------------------------------------------------------------------------------
    def norm2(self) -> float:
        return self._allreduce_sum([self.local.norm2()])[0]

    def _physical_index(self, logical_index: int) -> int:
        out = 0
        for lb in range(self.n):
            if (logical_index >> lb) & 1:
                out |= 1 << self.phys[lb]
        return out

    def probabilities(self, indices) -> np.ndarray:
        """|amplitude|^2 at the given (logical, reference-ordered) basis indices, on every rank."""
        import torch

        out = np.zeros(len(indices), dtype=np.float64)
        mine, where = [], []
        for j, idx in enumerate(indices):
            p = self._physical_index(int(idx))
            if p >> self.n_local == self.rank:
                mine.append(p & ((1 << self.n_local) - 1))
                where.append(j)
        if mine:
            out[where] = self.local.probabilities(mine)
        return np.array(self._allreduce_sum(out.tolist()), dtype=np.float64)

    def reduced_density(self, qubits) -> np.ndarray:
        """Reduced density matrix of ``qubits`` (at most six), on every rank: the kept qubits are made local (the
        trace over the rank bits is then a plain sum of the shards' matrices: one small all-reduce)."""
        qubits = [int(q) for q in qubits]
        if len(set(qubits)) != len(qubits):
            raise ValueError("Indices must be distinct.")
        for q in qubits:
            self._bit(q)
        self._localise_all(qubits, [])
        rho = self.local.reduced_density([self._local_qubit(self._bit(q)) for q in qubits])
        flat = self._allreduce_sum(np.concatenate([rho.real.ravel(), rho.imag.ravel()]).tolist())
        half = len(flat) // 2
        return (np.array(flat[:half]) + 1j * np.array(flat[half:])).reshape(rho.shape)

    def expect_pauli(self, paulis: str, qubits) -> complex:
        """``<psi| P |psi>`` for a Pauli string: X and Y legs are made local (they pair amplitudes), a Z on a rank bit
        is this rank's sign."""
        qubits = [int(q) for q in qubits]
        if len(paulis) != len(qubits):
            raise ValueError("one Pauli letter per qubit")
        for q in qubits:
            self._bit(q)
        flipping = [q for q, p in zip(qubits, paulis) if p in "XY"]
        self._localise_all(flipping, [q for q in qubits if q not in flipping])
        sign, letters, local = 1.0, [], []
        for q, p in zip(qubits, paulis):
            b = self._bit(q)
            if b >= self.n_local:
                if p == "Z" and self._rank_bit(b):
                    sign = -sign
                elif p not in "IZ":
                    raise ValueError(f"unknown Pauli letter {p!r}")
            else:
                letters.append(p)
                local.append(self._local_qubit(b))
        value = complex(self.local.expect_pauli("".join(letters), local)) if local else complex(self.local.norm2())
        re, im = self._allreduce_sum([sign * value.real, sign * value.imag])
        return complex(re, im)

    def to_numpy(self, root: int | None = None) -> np.ndarray | None:
        """The whole ket in reference order as a host array: on every rank (``root=None``; tests and small
        registers) or on rank ``root`` only (the others return ``None``).

        The shards travel one after the other in pieces of at most ``chunk_amps`` amplitudes and are written straight
        into the host array: no rank ever holds more than its own shard and one staging piece in HBM (the first form
        concatenated the whole register on every GPU -- 256 GiB at 34 qubits).  With ``root`` given only that rank
        allocates the 2^n host array and the pieces go point to point; otherwise every piece is a broadcast."""
        import torch

        self.local.sync()
        shard = 1 << self.n_local
        piece = min(shard, self.chunk_amps)
        keeps = root is None or self.rank == root
        physical = np.empty(1 << self.n, dtype=np.complex128) if keeps else None
        staging = []

        def stage():
            if not staging:
                staging.append(torch.empty(piece, dtype=self.buf.dtype, device=self.buf.device))
            return staging[0]

        for r in range(self.world):
            for c in range(shard // piece):
                mine = self.buf[c * piece:(c + 1) * piece]
                src = None
                if root is None:
                    src = mine if r == self.rank else stage()
                    if self.world > 1:
                        self._broadcast(src, r)
                elif r == root:
                    src = mine if self.rank == root else None
                elif self.rank == r:
                    self._p2p([(root, mine)], []).wait()
                elif self.rank == root:
                    src = stage()
                    self._p2p([], [(r, src)]).wait()
                if keeps and src is not None:
                    lo = r * shard + c * piece
                    physical[lo:lo + piece] = src.cpu().numpy()
        if not keeps:
            return None
        # physical[axis for bit p] -> logical: logical bit lb reads physical bit phys[lb]
        t = physical.reshape((2,) * self.n)                      # axis a <-> physical bit n-1-a
        axes = [self.n - 1 - self.phys[self.n - 1 - a] for a in range(self.n)]   # logical axis a <- physical axis
        if axes == list(range(self.n)):
            return physical
        return np.ascontiguousarray(t.transpose(axes)).reshape(-1)

    # ---- timing passthrough (bench.py) ------------------------------------------------------------
    def event_record(self, slot: int) -> None:
        self.local.event_record(slot)

    def event_elapsed_ms(self, a: int, b: int) -> float:
        return self.local.event_elapsed_ms(a, b)

    def last_kernel(self) -> str:
        return self.local.last_kernel()
