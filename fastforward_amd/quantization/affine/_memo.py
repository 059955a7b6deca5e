"""One-slot memo of A1 for activations that several quantizers with equal parameters quantize in a row.

In the reference's module graph q_proj / k_proj / v_proj (and gate_proj / up_proj) each quantize the SAME hidden state with
their own input quantizer (reference nn/linear.py:33; SURVEY 3.3: "q/k/v each quantize the same hidden state separately") —
after calibration those quantizers hold equal parameters, so the launches produce the same codes three (two) times. Here a
later quantizer whose parameters EQUAL an earlier one's reuses the earlier codes. Rules that keep this exact and free of hidden
synchronisation:

  * only plain HIP activation tensors (not Parameters: weights are re-quantized by their own quantizer, nn/linear.py:34)
    outside autograd; the slot is keyed on the tensor OBJECT (weak reference), its version counter, storage pointer, shape
    and dtype, so any write autograd can see, a new tensor, or a recycled allocation misses (this package's in-place kernels
    bump the version counters of what they write: ops.rope_, ops.add_rmsnorm_quantize(sum_inplace=True));
  * "equal parameters" means the same tensors at the same versions, or a verdict of ``torch.equal`` read ONCE on the host for
    a pair of parameter versions that had both been seen before — parameters a range estimator rewrites on every step never
    reach that point — and never while a hipGraph is being captured (then: no reuse, same result, one launch more);
  * the slot exists only INSIDE a ``sibling_quantizers()`` block — opened by this package's own module forwards around the
    sibling calls (q / k / v in QuantizedLlamaAttention, gate / up in QuantizedLlamaMLP, the same pairs in the fused forwards)
    and emptied when the outermost block ends. A version counter cannot see a write through ``tensor.data`` (or through a raw
    pointer of somebody else's kernel); between two sibling calls there is only this package's own code, so no such write can
    fall between a ``remember`` and its ``lookup``; everywhere else every quantizer call launches A1, as the reference does
    (nn/linear.py:32-39). Codes produced before a hipGraph capture began are never handed out during it (the graph would
    lack the A1 launch and replay stale codes) and the other way round; the slot is per thread.

While range estimators rewrite the parameters on every step none of that applies — no pair of versions is ever seen twice — and
the three (two) launches come back. ``sibling_quantizers(undecided=True)`` moves the question to the device for callers that
consume the codes through entry points which can ask it again: a later per-tensor int8 quantizer of the same tensor launches
``ops.quantize_by_tile_unless_same`` against the first sibling's parameters — nothing is read or written where they are the
same — and the QuantizedTensor it returns is MARKED with the first sibling's codes and parameters (``earlier_of``; the mark
lives on that object: a copy, view or re-wrap of it does not carry it). The opener of the block owns the consequences: every
reader of such codes inside the block is either ``ops.linear_w8a8_earlier`` / ``ops.mlp_gate_up_w8a8_estimating`` (which read
the codes in force) or is preceded by ``settle`` (which writes them, a ``torch.where`` on the device); marked tensors still
alive when the outermost block ends are settled there.
"""

from __future__ import annotations

import contextlib
import threading
import weakref

from typing import Any, Iterator

import torch


class RecentActivationCodes(threading.local):
    def __init__(self) -> None:
        self._data: weakref.ref | None = None
        self._key: tuple[Any, ...] = ()
        self._entries: list[tuple[Any, ...]] = []
        self._seen: dict[int, tuple[weakref.ref, int]] = {}  # parameter tensor -> version at its last sighting
        self._verdicts: dict[tuple[int, int, int, int], tuple[weakref.ref, weakref.ref, bool]] = {}
        self._depth = 0
        self.hits = 0
        self._extrema: tuple[weakref.ref, tuple[Any, ...], torch.Tensor] | None = None  # (data, key, [min, max] of it)
        self.extrema_hits = 0
        self._undecided = False
        self._marked: list[weakref.ref] = []  # QuantizedTensors whose codes the device may have left unwritten
        self.undecided_launches = 0

    @contextlib.contextmanager
    def scope(self, undecided: bool = False) -> Iterator[None]:
        """Sibling quantizer calls on one activation: reuse is allowed inside, the slot is emptied on the way out. `undecided`:
        the opener consumes the siblings' codes through entry points that take the earlier sibling's codes along (module
        docstring) — quantizers whose parameters cannot be compared on the host leave the comparison to the device."""
        self._depth += 1
        was = self._undecided
        self._undecided = was or undecided
        try:
            yield
        finally:
            self._depth -= 1
            self._undecided = was
            if self._depth == 0:
                self.clear()

    @staticmethod
    def _eligible(data: torch.Tensor, params: Any) -> bool:
        if type(data) is not torch.Tensor or not data.is_cuda:
            return False
        scale, offset = params.scale, params.offset
        if not isinstance(scale, torch.Tensor) or not (offset is None or isinstance(offset, torch.Tensor)):
            return False
        if torch.is_grad_enabled() and (data.requires_grad or scale.requires_grad or (offset is not None and offset.requires_grad)):
            return False
        return True

    def _stable(self, t: torch.Tensor | None) -> bool:
        """Seen before at this very version? (also records the sighting)"""
        if t is None:
            return True
        hit = self._seen.get(id(t))
        stable = hit is not None and hit[0]() is t and hit[1] == t._version
        if len(self._seen) > 8192:
            self._seen = {k: v for k, v in self._seen.items() if v[0]() is not None}
        self._seen[id(t)] = (weakref.ref(t), t._version)
        return stable

    def _same(self, a: torch.Tensor | None, b: torch.Tensor | None, stable: bool) -> bool:
        if a is None or b is None:
            return a is b
        if a is b:
            return True
        if a.shape != b.shape or a.dtype != b.dtype or a.device != b.device:
            return False
        key = (id(a), a._version, id(b), b._version)
        hit = self._verdicts.get(key)
        if hit is not None and hit[0]() is a and hit[1]() is b:
            return hit[2]
        if not stable or torch.cuda.is_current_stream_capturing():
            return False
        verdict = bool(torch.equal(a.detach(), b.detach()))  # one host read per pair of stable parameter versions
        if len(self._verdicts) > 8192:
            self._verdicts = {k: v for k, v in self._verdicts.items() if v[0]() is not None and v[1]() is not None}
        self._verdicts[key] = (weakref.ref(a), weakref.ref(b), verdict)
        return verdict

    @staticmethod
    def _data_key(data: torch.Tensor) -> tuple[Any, ...]:
        return (data._version, data.data_ptr(), tuple(data.shape), data.dtype, torch.cuda.is_current_stream_capturing())

    def lookup(self, data: torch.Tensor, params: Any, tile: Any, container: torch.dtype) -> torch.Tensor | None:
        if self._depth == 0 or not self._eligible(data, params):
            return None
        stable_scale, stable_offset = self._stable(params.scale), self._stable(params.offset)  # both sightings recorded
        if self._data is None or self._data() is not data or self._key != self._data_key(data):
            return None
        for scale, scale_v, offset, offset_v, bits, etile, econtainer, raw in self._entries:
            if bits != params.num_bits or econtainer != container or etile != tile:
                continue
            if scale._version != scale_v or (offset is not None and offset._version != offset_v):
                continue  # the earlier quantizer's parameters moved since: its codes are history
            if self._same(scale, params.scale, stable_scale and stable_offset) and self._same(offset, params.offset, stable_scale and stable_offset):
                self.hits += 1
                return raw
        return None

    def remember(self, data: torch.Tensor, params: Any, tile: Any, container: torch.dtype, raw: torch.Tensor) -> None:
        if self._depth == 0 or not self._eligible(data, params):
            return
        key = self._data_key(data)
        if self._data is None or self._data() is not data or self._key != key:
            self._data, self._key, self._entries = weakref.ref(data), key, []
        offset = params.offset
        self._entries.append((params.scale, params.scale._version, offset, -1 if offset is None else offset._version, params.num_bits, tile, container, raw))

    # Range ESTIMATION of siblings (RunningMinMax on q / k / v or gate / up inputs): every estimator needs the extrema of the
    # same activation; the reduction over the tensor is done once and each estimator merges the two numbers into its own
    # running state — exact whatever the states are (min / max are exact), unlike code reuse it needs no equal parameters.
    def extrema(self, data: torch.Tensor) -> torch.Tensor | None:
        """[min, max] of `data` (data dtype, on its device) left by an earlier sibling's estimator step, or None."""
        if self._depth == 0 or self._extrema is None or type(data) is not torch.Tensor:
            return None
        ref, key, pair = self._extrema
        if ref() is not data or key != self._data_key(data):
            return None
        self.extrema_hits += 1
        return pair

    def remember_extrema(self, data: torch.Tensor, pair: torch.Tensor) -> None:
        if self._depth > 0 and type(data) is torch.Tensor and data.is_cuda:
            self._extrema = (weakref.ref(data), self._data_key(data), pair)

    # Parameters the host cannot compare (module docstring, last paragraph)
    def earlier_for(self, data: torch.Tensor, params: Any, tile: Any, container: torch.dtype) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor | None] | None:
        """(codes, scale, offset) of the first sibling that quantized `data` the same way — per tensor, as many bits, into int8 —
        with parameters that have not moved since, for a quantizer whose own parameters `lookup` could not match."""
        if not self._undecided or container != torch.int8 or not self._eligible(data, params) or torch.is_grad_enabled():
            return None
        if self._data is None or self._data() is not data or self._key != self._data_key(data):
            return None

        def one_f32(t: torch.Tensor | None) -> bool:
            return t is None or (t.numel() == 1 and t.dtype == torch.float32 and t.device == data.device)

        if params.scale.numel() != 1 or not one_f32(params.scale) or not one_f32(params.offset):
            return None
        for scale, scale_v, offset, offset_v, bits, etile, econtainer, raw in self._entries:
            if bits != params.num_bits or econtainer != container or etile != tile or not one_f32(scale) or not one_f32(offset):
                continue
            if scale._version != scale_v or (offset is not None and offset._version != offset_v):
                continue
            return raw, scale, offset
        return None

    def mark_undecided(self, quantized: Any, earlier: tuple[torch.Tensor, torch.Tensor, torch.Tensor | None], scale: torch.Tensor, offset: torch.Tensor | None) -> None:
        """`quantized.raw_data` came from ``quantize_by_tile_unless_same(…, scale, offset, earlier's parameters)``: unwritten where
        they agree."""
        quantized._ffq_earlier = (earlier, scale, offset, self._versions(earlier[1], earlier[2], scale, offset))
        self._marked.append(weakref.ref(quantized))
        self.undecided_launches += 1

    @staticmethod
    def _versions(*tensors: torch.Tensor | None) -> tuple[int, ...]:
        return tuple(-1 if t is None else t._version for t in tensors)

    @classmethod
    def _checked(cls, mark: tuple[Any, ...]) -> None:
        """The launch compared the four parameter tensors as they were; whoever repeats the comparison must find them unchanged."""
        earlier, scale, offset, versions = mark
        if cls._versions(earlier[1], earlier[2], scale, offset) != versions:
            raise RuntimeError("quantization parameters were rewritten between a sibling quantizer's device-decided launch and the reader of its codes")

    @classmethod
    def earlier_of(cls, quantized: Any) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor | None] | None:
        mark = getattr(quantized, "_ffq_earlier", None)
        if mark is None:
            return None
        cls._checked(mark)
        return mark[0]

    @classmethod
    def settle(cls, quantized: Any) -> None:
        """Make `quantized` hold its quantizer's codes whatever the device decided (for a reader that cannot take the earlier codes
        along): the comparison of ``ffq_quantize_by_tile_unless_same`` once more, then a select — a pass over the codes, no host read."""
        mark = getattr(quantized, "_ffq_earlier", None)
        if mark is None:
            return
        del quantized._ffq_earlier
        cls._checked(mark)
        (codes, e_scale, e_offset), scale, offset, _ = mark
        raw = quantized.raw_data
        zero = torch.zeros((), dtype=torch.float32, device=raw.device)
        o = zero if offset is None else torch.round(offset.detach().reshape(()))
        eo = zero if e_offset is None else torch.round(e_offset.detach().reshape(()))
        same = (scale.detach().reshape(()).view(torch.int32) == e_scale.detach().reshape(()).view(torch.int32)) & (o == eo)
        raw.copy_(torch.where(same, codes, raw))

    @property
    def inside_scope(self) -> bool:
        return self._depth > 0

    def clear(self) -> None:
        self._data, self._key, self._entries = None, (), []
        self._extrema = None
        marked, self._marked = self._marked, []
        # codes that outlive the block are ordinary codes from here on. `clear` runs from the `finally` of `scope()`: one tensor whose
        # parameters were rewritten (settle -> _checked raises) must neither leave the REST with unwritten codes nor mask the exception
        # that is unwinding — every marked tensor is settled, the first failure is raised behind the loop
        failure: RuntimeError | None = None
        for ref in marked:
            quantized = ref()
            if quantized is None:
                continue
            try:
                self.settle(quantized)
            except RuntimeError as e:
                failure = failure or e
        if failure is not None:
            raise failure


RECENT = RecentActivationCodes()
sibling_quantizers = RECENT.scope
