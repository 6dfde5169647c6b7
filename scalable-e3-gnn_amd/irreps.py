"""Irreps bookkeeping for the L1 tensor-product hot path.

The reference uses ``e3nn.o3.Irreps`` / ``e3nn.o3.Instruction`` only for bookkeeping
(`/root/reference/models/segnn/l1_tensor_prod.py:5,13-21,29-36,44-51,58-65,122-151,193`):
``.lmax``, ``.dim``, ``len()``, iteration yielding entries with ``.mul``, ``.dim``, ``.ir.l``,
``.ir.p``, ``.ir.dim``, ``Irreps.spherical_harmonics(1)`` and the ``Instruction`` named tuple.
e3nn is not a dependency of this repo; this module provides exactly that surface.  Real
``e3nn`` objects are accepted everywhere through duck typing (``as_blocks``).

Semantics kept from e3nn: entry order is preserved (no sorting / simplification), parity
``p`` is ``+1`` (``e``) or ``-1`` (``o``), ``Irrep.dim = 2l+1``, spherical-harmonics parity is
``p**l`` with ``p=-1``.
"""
from __future__ import annotations

import re
from collections import namedtuple
from typing import Iterable, Iterator, List, Sequence, Tuple, Union

# e3nn-compatible field order (used positionally by the reference: `L1TP.py:151,193`).
Instruction = namedtuple(
    "Instruction",
    ["i_in1", "i_in2", "i_out", "connection_mode", "has_weight", "path_weight", "path_shape"],
)


class Irrep:
    """One irreducible representation of O(3): degree ``l`` and parity ``p`` (+1 / -1)."""

    __slots__ = ("l", "p")

    def __init__(self, l: Union[int, str, "Irrep"], p: int | None = None):
        if p is None:
            if isinstance(l, Irrep):
                l, p = l.l, l.p
            elif isinstance(l, str):
                m = re.fullmatch(r"\s*(\d+)([eoy])\s*", l)
                if m is None:
                    raise ValueError(f"unable to convert string {l!r} into an Irrep")
                deg = int(m.group(1))
                p = {"e": 1, "o": -1, "y": (-1) ** deg}[m.group(2)]
                l = deg
            elif hasattr(l, "l") and hasattr(l, "p"):  # e3nn.o3.Irrep
                l, p = int(l.l), int(l.p)
            else:
                l, p = l  # tuple
        if int(l) < 0 or int(p) not in (-1, 1):
            raise ValueError(f"invalid irrep l={l}, p={p}")
        self.l = int(l)
        self.p = int(p)

    @property
    def dim(self) -> int:
        return 2 * self.l + 1

    def __iter__(self):
        yield self.l
        yield self.p

    def __eq__(self, other) -> bool:
        try:
            o = Irrep(other)
        except Exception:
            return NotImplemented
        return (self.l, self.p) == (o.l, o.p)

    def __hash__(self) -> int:
        return hash((self.l, self.p))

    def __repr__(self) -> str:
        return f"{self.l}{'e' if self.p == 1 else 'o'}"


class MulIr:
    """``mul`` copies of an :class:`Irrep` — what iterating an :class:`Irreps` yields."""

    __slots__ = ("mul", "ir")

    def __init__(self, mul: int, ir: Irrep):
        self.mul = int(mul)
        self.ir = ir

    @property
    def dim(self) -> int:
        return self.mul * self.ir.dim

    def __iter__(self):
        yield self.mul
        yield self.ir

    def __len__(self) -> int:
        return 2

    def __getitem__(self, i):
        return (self.mul, self.ir)[i]

    def __eq__(self, other) -> bool:
        try:
            return self.mul == other.mul and self.ir == other.ir
        except AttributeError:
            return NotImplemented

    def __hash__(self) -> int:
        return hash((self.mul, self.ir))

    def __repr__(self) -> str:
        return f"{self.mul}x{self.ir}"


class Irreps:
    """Ordered direct sum of ``mul x irrep`` blocks, e.g. ``Irreps("8x0e+8x1o")``."""

    def __init__(self, spec: Union[str, "Irreps", Iterable, None] = None):
        blocks: List[MulIr] = []
        if spec is None:
            pass
        elif isinstance(spec, Irreps):
            blocks = list(spec._blocks)
        elif isinstance(spec, str):
            text = spec.strip()
            if text:
                for term in text.split("+"):
                    term = term.strip()
                    if "x" in term:
                        mul_s, ir_s = term.split("x", 1)
                        mul = int(mul_s)
                    else:
                        mul, ir_s = 1, term
                    if mul < 0:
                        raise ValueError(f"negative multiplicity in {spec!r}")
                    blocks.append(MulIr(mul, Irrep(ir_s)))
        elif isinstance(spec, Irrep):
            blocks = [MulIr(1, spec)]
        else:
            for entry in spec:  # e3nn.o3.Irreps, list of (mul, ir), list of MulIr ...
                if hasattr(entry, "mul") and hasattr(entry, "ir"):
                    blocks.append(MulIr(entry.mul, Irrep(entry.ir)))
                elif isinstance(entry, (str, Irrep)):
                    blocks.append(MulIr(1, Irrep(entry)))
                else:
                    mul, ir = entry
                    blocks.append(MulIr(mul, Irrep(ir)))
        self._blocks: Tuple[MulIr, ...] = tuple(blocks)

    # -- e3nn surface used by the reference ---------------------------------------------------
    @staticmethod
    def spherical_harmonics(lmax: int, p: int = -1) -> "Irreps":
        return Irreps([(1, (l, p ** l)) for l in range(lmax + 1)])

    @property
    def dim(self) -> int:
        return sum(b.dim for b in self._blocks)

    @property
    def num_irreps(self) -> int:
        return sum(b.mul for b in self._blocks)

    @property
    def lmax(self) -> int:
        if not self._blocks:
            raise ValueError("Cannot get lmax of empty Irreps")
        return max(b.ir.l for b in self._blocks)

    @property
    def ls(self) -> List[int]:
        return [b.ir.l for b in self._blocks for _ in range(b.mul)]

    def __iter__(self) -> Iterator[MulIr]:
        return iter(self._blocks)

    def __len__(self) -> int:
        return len(self._blocks)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return Irreps(self._blocks[i])
        return self._blocks[i]

    def __add__(self, other) -> "Irreps":
        return Irreps(list(self._blocks) + list(Irreps(other)._blocks))

    def __radd__(self, other) -> "Irreps":
        return Irreps(other) + self

    def __eq__(self, other) -> bool:
        try:
            return self._blocks == Irreps(other)._blocks
        except Exception:
            return NotImplemented

    def __hash__(self) -> int:
        return hash(self._blocks)

    def __repr__(self) -> str:
        return "+".join(repr(b) for b in self._blocks)

    def slices(self) -> List[slice]:
        out, i = [], 0
        for b in self._blocks:
            out.append(slice(i, i + b.dim))
            i += b.dim
        return out


def as_blocks(irreps) -> List[Tuple[int, int, int]]:
    """``[(l, p, mul), ...]`` for our :class:`Irreps`, a string, or a real ``e3nn.o3.Irreps``."""
    if isinstance(irreps, str):
        irreps = Irreps(irreps)
    if isinstance(irreps, (list, tuple)) and all(isinstance(b, (list, tuple)) and len(b) == 3 for b in irreps):
        return [(int(l), int(p), int(mul)) for l, p, mul in irreps]  # already blocks
    return [(int(m.ir.l), int(m.ir.p), int(m.mul)) for m in irreps]


def irreps_dim(irreps) -> int:
    return sum((2 * l + 1) * mul for l, _, mul in as_blocks(irreps))
