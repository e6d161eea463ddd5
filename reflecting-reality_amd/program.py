"""Step programs: the C-ABI call sequence of one forward pass, exported so that a host WITHOUT Python can run it.

SURVEY.md §8(b) asks for step-level C entries (`mf_brushnet_forward`, `mf_unet_forward`, `mf_denoise_step_fused`) next to the
per-operator ones.  The sequencing of a step (which launches, on which buffers, with which tiles) lives in models.py /
pipeline.py; instead of restating it in C++, the library replays it: a `Recorder` runs one eager pass of any function built
on hip.py (BrushNet, the UNet, the whole denoise step of pipeline_brushnet.py:1250-1332, the VAE) and writes down every
`mf_*` launch with its descriptor, every device copy torch made in between (as `mf_memcpy2d` / `mf_memset`), and the
device buffers they touch:

  * io buffers      — tensors the caller names ("latents", "coef4", ...): the host binds its own memory to them,
  * constants       — everything else that was alive before the pass (weights in their device layouts, the cross-attention
                      K / V^T of the bound prompt, time-embedding tables, scratch): their bytes are written into the file,
  * workspace       — the caching allocator's segments of a private pool the pass allocated from: one region per segment,
                      so the recorded addresses (including the allocator's reuse of freed blocks) stay valid as offsets.

`mf_program_load` / `mf_program_bind` / `mf_program_run` (csrc/program.cpp, include/mfhip.h) replay the file through the same
entry points, on any stream, capturable into a hipGraph by the host.  A program is specialised like a hipGraph: shapes, precision,
tiles and scalar arguments (guidance scale, conditioning scale) are the recorded ones.  Anything the recorder cannot express (a
torch kernel other than a copy / fill, an entry without a replay thunk) raises at record time — there is no partial export.
"""
import ctypes as C
import struct
from typing import Dict, List, Optional, Tuple

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from . import hip

MAGIC = b"MFPROG1\0"
KIND_CONST, KIND_WORKSPACE, KIND_IO = 0, 1, 2
A_I32, A_I64, A_F32, A_PTR, A_DESC = 0, 1, 2, 3, 4

# replayable entries: one character per argument before the trailing stream — p device pointer, i int32, l int64, f float,
# d descriptor struct (csrc/program.cpp holds the matching thunk for each; tests/test_program_gpu.py replays every one of them)
SIGNATURES = {
    "mf_gemm_conv": "d", "mf_groupnorm": "d",
    "mf_layernorm": "pipipplif", "mf_softmax_rows": "ppilii",
    "mf_attention_bf16": "plplplpliiiiif", "mf_attention_f16": "plplplpliiiiif",
    "mf_attention_f16x3": "pplpplpplpliiiiif", "mf_attention_f16x3_lse": "pplpplpplplpiiiiif", "mf_split_halves": "pppl",
    "mf_quantize_rows_fp8": "pipplippf",
    "mf_pack_nhwc": "pipipiiii", "mf_unpack_nchw": "pilpiii",
    "mf_add": "pipipil", "mf_cast_bf16": "ppl", "mf_geglu": "pipili",
    "mf_timestep_embedding": "ppiiif", "mf_silu_f32": "ppl",
    "mf_cfg_ddim_step_dev": "ppfpppifl", "mf_cfg_combine": "ppfpl",
    "mf_vae_sample": "pilppiiif", "mf_nearest_resize": "ppiiiii",
    "mf_transpose": "ppiiillll", "mf_transpose_bf16": "ppiiillll", "mf_transpose_bf16_bf16": "ppiiillll",
    "mf_memcpy2d": "plplll", "mf_memset": "pil",
}
# queries / developer switches: forwarded, never recorded
_PASS_THROUGH = ("mf_last_error", "mf_abi_version", "mf_gemm_num_tiles", "mf_gemm_tile_shape", "mf_gemm_tile_table_version",
                 "mf_groupnorm_ws_floats", "mf_sizeof_gemm_desc", "mf_sizeof_groupnorm_desc")
# descriptor fields that are HOST out-pointers (the library reports a choice through them): null in a program
_HOST_FIELDS = {"gn_part_rows", "gn_grouped", "deferred_splits"}

# aten ops that launch nothing (views, allocations, metadata)
_VIEW_OPS = {
    "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "view", "_unsafe_view", "reshape", "slice", "select", "narrow",
    "as_strided", "detach", "alias", "t", "transpose", "permute", "unsqueeze", "squeeze", "expand", "split", "split_with_sizes",
    "chunk", "unbind", "view_as", "flatten", "unflatten", "_reshape_alias", "lift_fresh", "is_contiguous", "size", "stride",
    "storage_offset", "numel", "dim", "sym_size", "sym_stride", "sym_numel", "sym_storage_offset", "is_pinned",
    "record_stream",            # an allocator note (which stream consumes the block), not a launch
}


class ProgramError(hip.MfhipError):
    pass


def _rows_runs(t: torch.Tensor) -> Tuple[int, int, int]:
    """A strided tensor as `height` runs of `run` contiguous elements `pitch` elements apart (what a 2-D copy moves), or raise."""
    sizes = [s for s in t.shape if s != 1]
    strides = [st for s, st in zip(t.shape, t.stride()) if s != 1]
    if not sizes:
        return 1, 1, 1
    run, i = 1, len(sizes) - 1
    while i >= 0 and strides[i] == run:
        run *= sizes[i]
        i -= 1
    if i < 0:
        return 1, run, run
    height, pitch = 1, strides[i]
    expect = pitch
    while i >= 0:
        if strides[i] != expect:
            raise ProgramError(f"a torch copy of a tensor with sizes {tuple(t.shape)} / strides {tuple(t.stride())} is not a 2-D copy")
        height *= sizes[i]
        expect *= sizes[i]
        i -= 1
    return height, pitch, run


def _dense_bytes(name: str, t: torch.Tensor) -> int:
    """Bytes of the dense block a tensor covers (any permutation of a contiguous tensor, e.g. a channels-last view), or raise."""
    extent = 1 + sum((s - 1) * st for s, st in zip(t.shape, t.stride())) if t.numel() else 0
    if not t.is_cuda or extent != t.numel():
        raise ProgramError(f"buffer {name!r}: a device tensor that covers one dense block of memory is needed "
                           f"(sizes {tuple(t.shape)}, strides {tuple(t.stride())}, {t.device})")
    return extent * t.element_size()


class _Proxy:
    """Stands in for the ctypes library while a Recorder is active: every launch entry is written down, then forwarded."""

    def __init__(self, rec: "Recorder", lib: C.CDLL):
        self._rec, self._lib = rec, lib

    def __getattr__(self, name: str):
        fn = getattr(self._lib, name)
        if name in _PASS_THROUGH:
            return fn
        if name not in SIGNATURES:
            def refuse(*a, **k):
                raise ProgramError(f"{name} has no replay thunk (program.SIGNATURES / csrc/program.cpp): it cannot be part of an exported program")
            return refuse
        rec = self._rec

        def call(*args):
            rec._record(name, args)
            return fn(*args)
        return call


class _TorchOps(TorchDispatchMode):
    """torch kernels between the launches: copies and fills are recorded as mf_memcpy2d / mf_memset, anything else is refused."""

    def __init__(self, rec: "Recorder"):
        super().__init__()
        self.rec = rec

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = func.overloadpacket.__name__
        out = func(*args, **kwargs)
        if name in _VIEW_OPS:
            return out
        flat = [a for a in list(args) + list(kwargs.values()) if isinstance(a, torch.Tensor)]
        for a in args:
            if isinstance(a, (list, tuple)):
                flat += [x for x in a if isinstance(x, torch.Tensor)]
        outs = [out] if isinstance(out, torch.Tensor) else [o for o in out if isinstance(o, torch.Tensor)] if isinstance(out, (list, tuple)) else []
        if not any(t.is_cuda for t in flat + outs):
            return out
        rec = self.rec
        if name == "copy_":
            rec._copy(args[0], args[1])
        elif name in ("clone", "contiguous"):
            if out.data_ptr() != args[0].data_ptr():
                rec._copy(out, args[0])
        elif name == "_to_copy":
            src = args[0]
            if not src.is_cuda or out.dtype != src.dtype:
                raise ProgramError(f"a torch conversion {src.dtype} ({src.device}) -> {out.dtype} inside the recorded pass")
            rec._copy(out, src)
        elif name == "cat":
            dim = args[1] if len(args) > 1 else kwargs.get("dim", 0)
            rec._cat(out, list(args[0]), dim % out.dim())
        elif name in ("zero_", "zeros", "zeros_like", "new_zeros"):
            rec._fill(out, 0)
        elif name == "fill_" and float(args[1]) == 0.0:
            rec._fill(out, 0)
        else:
            raise ProgramError(f"torch op aten::{name} launches a kernel inside the recorded pass: replace it by an mf_* entry "
                               f"(tensors {[tuple(t.shape) for t in flat]})")
        return out


class Recorder:
    """with Recorder(named) as rec: <one eager pass>   then   <restore the inputs>; rec.save(path)"""

    def __init__(self, named: Dict[str, torch.Tensor], tables: Optional[Dict[str, torch.Tensor]] = None, device=None, capture: bool = False):
        """`named`: the io buffers.  `tables`: device tensors stored in the file as named constants although no launch reads them
        (per-step rows the host copies into an io buffer between runs: DDIM coefficients, time-embedding tables).
        `capture`: the pass is recorded INSIDE a torch.cuda.graph capture (create the Recorder before the capture begins, enter it
        inside, call finish(graph.pool()) after the capture ended).  Only this mode takes a pass on several streams: forks and
        joins (Stream.wait_stream / wait_event, Event.record) become part of the program, and the workspace is the graph's private
        pool — whose reuse of freed blocks is ordered by the streams' dependencies, not by what had finished on the clock, as an
        eager pass's would be."""
        self.device = torch.device(device if device is not None else next(iter(named.values())).device)
        self.named = {k: v for k, v in named.items()}
        self.tables = dict(tables or {})
        for k, v in list(self.named.items()) + list(self.tables.items()):
            _dense_bytes(k, v)
        self.calls: List[Tuple[str, list]] = []
        self.outputs: Dict[str, Tuple[int, int]] = {}
        self._keep: List[torch.Tensor] = []
        self.layouts: Dict[str, dict] = {}
        self._before = None
        self._pool = None
        self._ctx = []
        self.capture = bool(capture)
        self.error = None
        self._streams: Dict[int, int] = {}
        self._events: Dict[int, int] = {}
        self._patched = []
        if self.capture:
            torch.cuda.synchronize(self.device)
            self._before = torch.cuda.memory_snapshot()

    # ---- recording -------------------------------------------------------------------------------------------------------
    def __enter__(self):
        if hip._RECORDER is not None:
            raise ProgramError("a Recorder is already active")
        lib = hip.load()
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing != self.capture:
            raise ProgramError("Recorder(capture=True) is entered inside a torch.cuda.graph capture, a plain Recorder outside one")
        self._streams = {torch.cuda.current_stream(self.device).cuda_stream: 0}
        if self.capture:
            ops_ctx = _TorchOps(self)
            ops_ctx.__enter__()
            self._ctx = [ops_ctx]
            self._patch_syncs()
        else:
            torch.cuda.synchronize(self.device)
            self._before = torch.cuda.memory_snapshot()
            self._pool = torch.cuda.MemPool()
            self._enter_contexts()
        hip._RECORDER = _Proxy(self, lib)
        return self

    def __exit__(self, *exc):
        hip._RECORDER = None
        self.error = exc[1] if isinstance(exc[1], ProgramError) else None      # (a capture that ends on an exception reports its own error on top)
        for obj, name, orig in self._patched:
            setattr(obj, name, orig)
        self._patched = []
        for c in self._ctx:
            c.__exit__(*exc)
        self._ctx = []
        if not self.capture:
            torch.cuda.synchronize(self.device)
            if exc[0] is None:
                self._resolve(tuple(self._pool.id))
        return False

    def finish(self, pool_id) -> None:
        """capture mode: after the capture ended — `pool_id` = graph.pool(), the private pool the captured pass allocated from"""
        if not self.capture:
            raise ProgramError("finish() belongs to Recorder(capture=True)")
        self._resolve(tuple(pool_id))

    # ---- streams (capture mode) ----------------------------------------------------------------------------------------------
    def _sid(self, handle: int) -> int:
        handle = int(handle or 0)
        if handle not in self._streams:
            if not self.capture:
                raise ProgramError("a launch on another stream than the recording one: an eager pass is recorded on ONE stream "
                                   "(Recorder(capture=True) takes forks and joins)")
            self._streams[handle] = len(self._streams)
        return self._streams[handle]

    def _eid(self, ev) -> int:
        return self._events.setdefault(id(ev), len(self._events))

    def _patch_syncs(self) -> None:
        rec = self
        E = torch.cuda.Event
        # (torch's Stream.wait_stream / wait_event / record_event all end in Event.record and Event.wait: two patches see every sync)
        o_record, o_ewait = E.record, E.wait

        def record(self_e, stream=None):
            st = stream if stream is not None else torch.cuda.current_stream()
            rec.calls.append(("@record", [(A_I32, rec._eid(self_e))], rec._sid(st.cuda_stream)))
            return o_record(self_e, st)

        def ewait(self_e, stream=None):
            st = stream if stream is not None else torch.cuda.current_stream()
            rec.calls.append(("@wait", [(A_I32, rec._eid(self_e))], rec._sid(st.cuda_stream)))
            return o_ewait(self_e, st)
        for obj, name, new, orig in ((E, "record", record, o_record), (E, "wait", ewait, o_ewait)):
            setattr(obj, name, new)
            self._patched.append((obj, name, orig))

    def output(self, name: str, t: torch.Tensor) -> None:
        """Name a tensor the pass PRODUCED (call inside the `with` block): the program ends with a copy of it into an io buffer of
        its own.  (The tensor's own memory cannot be the io buffer: the caching allocator may have lent the same addresses to an
        earlier, larger intermediate of the pass, whose launches would then write past the end of the host's buffer.  The copy's
        destination is allocated OUTSIDE the pass's private pool, where no recorded launch can have been.)  The layout — a dense
        permutation, e.g. the channels-last views the models return — goes to `layouts`."""
        nbytes = _dense_bytes(name, t)
        if self.capture:
            raise ProgramError("output() is for eager recordings; a captured pass writes its results into named buffers")
        for c in self._ctx:
            c.__exit__(None, None, None)
        o = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device)
        o.copy_(t)
        self._enter_contexts()
        self._keep.append(o)
        if nbytes:
            self._mem("mf_memcpy2d", o.data_ptr(), nbytes, t.data_ptr(), nbytes, nbytes, 1)
        self.outputs[name] = (o.data_ptr(), nbytes)
        self.layouts[name] = dict(shape=list(t.shape), strides=list(t.stride()), dtype=str(t.dtype).replace("torch.", ""))

    def _enter_contexts(self) -> None:
        pool_ctx = torch.cuda.use_mem_pool(self._pool, device=self.device)
        ops_ctx = _TorchOps(self)
        pool_ctx.__enter__()
        ops_ctx.__enter__()
        self._ctx = [ops_ctx, pool_ctx]

    def _record(self, name: str, args: tuple) -> None:
        sig = SIGNATURES[name]
        if len(args) != len(sig) + 1:
            raise ProgramError(f"{name}: {len(args)} arguments for signature {sig!r} + stream")
        st = args[-1]
        st = st.value if isinstance(st, C.c_void_p) else st
        sid = self._sid(st)
        rec = []
        for kind, a in zip(sig, args[:-1]):
            if kind == "p":
                v = a.value if isinstance(a, C.c_void_p) else a
                if v is not None and not isinstance(v, int):
                    raise ProgramError(f"{name}: pointer argument of type {type(a).__name__}")
                rec.append((A_PTR, int(v or 0)))
            elif kind == "i":
                rec.append((A_I32, int(a.value if hasattr(a, "value") else a)))
            elif kind == "l":
                rec.append((A_I64, int(a.value if hasattr(a, "value") else a)))
            elif kind == "f":
                rec.append((A_F32, float(a.value if hasattr(a, "value") else a)))
            else:
                obj = getattr(a, "_obj", a)
                if not isinstance(obj, C.Structure):
                    raise ProgramError(f"{name}: descriptor argument of type {type(a).__name__}")
                raw = bytearray(bytes(obj))
                fix = []
                for fname, ftype in obj._fields_:
                    if ftype is not C.c_void_p:
                        continue
                    off = getattr(type(obj), fname).offset
                    val = getattr(obj, fname) or 0
                    raw[off:off + 8] = b"\0" * 8
                    if val and fname not in _HOST_FIELDS:
                        fix.append((off, int(val)))
                rec.append((A_DESC, bytes(raw), fix))
        self.calls.append((name, rec, sid))

    def _mem(self, name: str, *vals) -> None:
        sig = SIGNATURES[name]
        kinds = {"p": A_PTR, "i": A_I32, "l": A_I64}
        sid = self._sid(torch.cuda.current_stream(self.device).cuda_stream)
        self.calls.append((name, [(kinds[k], int(v)) for k, v in zip(sig, vals)], sid))

    def _copy(self, dst: torch.Tensor, src: torch.Tensor) -> None:
        if not (dst.is_cuda and src.is_cuda):
            raise ProgramError("a host <-> device copy inside the recorded pass: the program's inputs are device buffers")
        if dst.dtype != src.dtype:
            raise ProgramError(f"a converting torch copy ({src.dtype} -> {dst.dtype}) inside the recorded pass")
        if dst.numel() == 0:
            return
        src = src.expand_as(dst) if src.shape != dst.shape else src
        es = dst.element_size()
        hd, pd, rd = _rows_runs(dst)
        hs, ps, rs = _rows_runs(src)
        if hd == 1 and hs == 1:
            self._mem("mf_memcpy2d", dst.data_ptr(), rd * es, src.data_ptr(), rd * es, rd * es, 1)
            return
        run = min(rd, rs)
        if rd % run or rs % run:
            raise ProgramError("a torch copy whose two sides have incompatible contiguous runs")
        # split the side with the longer run into rows of the shorter one
        def rows(h, p, r):
            return (h * (r // run), run if r > run else p) if (h == 1 or r == run) else None
        d2, s2 = rows(hd, pd, rd), rows(hs, ps, rs)
        if d2 is None or s2 is None or d2[0] != s2[0]:
            raise ProgramError(f"a torch copy {tuple(src.shape)}/{tuple(src.stride())} -> {tuple(dst.shape)}/{tuple(dst.stride())} is not a 2-D copy")
        self._mem("mf_memcpy2d", dst.data_ptr(), d2[1] * es, src.data_ptr(), s2[1] * es, run * es, d2[0])

    def _cat(self, out: torch.Tensor, parts: List[torch.Tensor], dim: int) -> None:
        if not out.is_contiguous():
            raise ProgramError("torch.cat into a non-contiguous tensor")
        es = out.element_size()
        outer = 1
        for s in out.shape[:dim]:
            outer *= s
        inner = 1
        for s in out.shape[dim + 1:]:
            inner *= s
        dpitch = out.shape[dim] * inner * es
        off = 0
        for p in parts:
            if p.numel() == 0:
                continue
            if not p.is_contiguous() or p.dtype != out.dtype:
                raise ProgramError("torch.cat of non-contiguous or mixed-dtype tensors inside the recorded pass")
            w = p.shape[dim] * inner * es
            self._mem("mf_memcpy2d", out.data_ptr() + off, dpitch, p.data_ptr(), w, w, outer)
            off += w

    def _fill(self, t: torch.Tensor, value: int) -> None:
        if not t.is_contiguous():
            raise ProgramError("a fill of a non-contiguous tensor inside the recorded pass")
        if t.numel():
            self._mem("mf_memset", t.data_ptr(), value, t.numel() * t.element_size())

    # ---- address -> (buffer, offset) ---------------------------------------------------------------------------------------
    def _resolve(self, pool_id: tuple) -> None:
        dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        after = torch.cuda.memory_snapshot()
        self.buffers: List[dict] = []            # kind, name, addr, bytes
        index: Dict[Tuple[int, int], int] = {}

        def add(kind, name, addr, nbytes):
            key = (addr, kind)
            if key not in index:
                index[key] = len(self.buffers)
                self.buffers.append(dict(kind=kind, name=name, addr=addr, bytes=nbytes))
            return index[key]
        io = [(k, v.data_ptr(), _dense_bytes(k, v)) for k, v in self.named.items()] + [(k, a, n) for k, (a, n) in self.outputs.items()]
        for k, v in self.named.items():
            self.layouts.setdefault(k, dict(shape=list(v.shape), strides=list(v.stride()), dtype=str(v.dtype).replace("torch.", "")))
        for k, a, n in io:
            add(KIND_IO, k, a, n)
        segs = [(s["address"], s["total_size"]) for s in after if s["device"] == dev and tuple(s.get("segment_pool_id", (0, 0))) == pool_id]
        blocks = []
        for s in self._before:
            if s["device"] != dev:
                continue
            a = s["address"]
            for b in s["blocks"]:
                if b["state"] == "active_allocated":
                    blocks.append((a, b["size"]))
                a += b["size"]
        blocks.sort()
        import bisect
        starts = [b[0] for b in blocks]

        def where(addr: int) -> Tuple[int, int]:
            for k, a, n in io:
                if a <= addr < a + n:
                    return index[(a, KIND_IO)], addr - a
            for a, n in segs:
                if a <= addr < a + n:
                    return add(KIND_WORKSPACE, f"workspace.{a:x}", a, n), addr - a
            i = bisect.bisect_right(starts, addr) - 1
            if i >= 0 and blocks[i][0] <= addr < blocks[i][0] + blocks[i][1]:
                return add(KIND_CONST, f"const.{blocks[i][0]:x}", blocks[i][0], blocks[i][1]), addr - blocks[i][0]
            raise ProgramError(f"device address {addr:#x} of a recorded launch is neither an io buffer, nor alive before the pass, nor allocated by it")
        out = []
        for name, rec, sid in self.calls:
            args = []
            for a in rec:
                if a[0] == A_PTR:
                    args.append((A_PTR,) + (where(a[1]) if a[1] else (-1, 0)))
                elif a[0] == A_DESC:
                    args.append((A_DESC, a[1], [(off,) + where(v) for off, v in a[2]]))
                else:
                    args.append(a)
            out.append((name, args, sid))
        self.resolved = out
        for k, v in self.tables.items():
            self.buffers.append(dict(kind=KIND_CONST, name=k, addr=v.data_ptr(), bytes=v.numel() * v.element_size()))

    # ---- the file -------------------------------------------------------------------------------------------------------
    def save(self, path: str, meta: str = "") -> dict:
        """Write the program; constants and io buffers carry the bytes they hold NOW (restore the pass's inputs first)."""
        header, placed, total, nstreams, nevents = serialize_header(self.resolved, self.buffers, len(self._events), meta)
        torch.cuda.synchronize(self.device)
        with open(path, "wb") as f:
            f.write(header)
            for off, b in placed:
                f.seek(off)
                host = torch.empty(b["bytes"], dtype=torch.uint8)
                hip_memcpy_d2h(host, b["addr"], b["bytes"])
                f.write(host.numpy().tobytes())
            f.truncate(total)
        return dict(calls=len(self.resolved), buffers=len(self.buffers), bytes=total,
                    const_bytes=sum(b["bytes"] for b in self.buffers if b["kind"] == KIND_CONST),
                    workspace_bytes=sum(b["bytes"] for b in self.buffers if b["kind"] == KIND_WORKSPACE),
                    streams=nstreams, events=nevents, entries=sorted({c[0] for c in self.resolved}))


def serialize_header(resolved, buffers, nevents: int, meta: str = ""):
    """The part of a program file mf_program_load parses (csrc/program.hip): magic, ABI version, counts, header length, meta, the buffer
    table with the file offsets of the buffers that carry data, then every call.  `resolved`: (entry, args, stream) with args
    (A_I32 | A_I64, value), (A_F32, value), (A_PTR, buffer, offset), (A_DESC, bytes, [(field offset, buffer, offset)]); `buffers`: dicts
    with kind / name / bytes.  Returns (header bytes, [(file offset, buffer)] to fill in, file length, streams, events)."""
    pad8 = lambda b: b + b"\0" * (-len(b) % 8)
    body = bytearray()
    for name, args, sid in resolved:
        nb = name.encode()
        body += struct.pack("<IIII", len(nb), len(args), sid, 0) + pad8(nb)
        tail = bytearray()
        for a in args:
            if a[0] == A_PTR:
                body += struct.pack("<Iiq", A_PTR, a[1], a[2])
            elif a[0] == A_DESC:
                body += struct.pack("<Iiq", A_DESC, len(a[2]), len(a[1]))
                tail += pad8(a[1])
                for off, buf, boff in a[2]:
                    tail += struct.pack("<Iiq", off, buf, boff)
            elif a[0] == A_F32:
                body += struct.pack("<Iifi", A_F32, 0, a[1], 0)
            else:
                body += struct.pack("<Iiq", a[0], 0, a[1])
        body += tail
    metab = pad8(meta.encode())
    table_len = sum(24 + len(pad8(b["name"].encode())) for b in buffers)
    head_len = 8 + 4 * 4 + 8 + 8 + 8 + len(metab) + table_len + len(body)
    cursor = (head_len + 255) // 256 * 256
    table = bytearray()
    placed = []
    for b in buffers:
        nb = b["name"].encode()
        if b["kind"] == KIND_WORKSPACE:
            off = -1
        else:
            off = cursor
            cursor = (cursor + b["bytes"] + 255) // 256 * 256
            placed.append((off, b))
        table += struct.pack("<IIqq", b["kind"], len(nb), b["bytes"], off) + pad8(nb)
    nstreams = max([c[2] for c in resolved] + [0]) + 1
    head = MAGIC + struct.pack("<IIII", hip.ABI_VERSION, len(buffers), len(resolved), len(metab)) + struct.pack("<qqII", head_len, cursor, nstreams, nevents) + metab
    out = bytes(head + table + body)
    assert len(out) == head_len
    return out, placed, cursor, nstreams, nevents


def hip_memcpy_d2h(host: torch.Tensor, addr: int, nbytes: int) -> None:
    """Device bytes at a raw address -> a host uint8 tensor (the runtime's hipMemcpy; the address need not start a torch tensor)."""
    rt = _hiprt()
    rc = rt.hipMemcpy(C.c_void_p(host.data_ptr()), C.c_void_p(addr), C.c_size_t(nbytes), 2)
    if rc != 0:
        raise ProgramError(f"hipMemcpy(device -> host, {nbytes} bytes at {addr:#x}) failed: {rc}")


_rt = None


def _hiprt():
    global _rt
    if _rt is None:
        import os
        tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        _rt = C.CDLL(tl if os.path.exists(tl) else "libamdhip64.so")
    return _rt


class Program:
    """A loaded program with torch-owned memory behind every buffer (tests, and Python hosts that want the replay without the models)."""

    def __init__(self, path: str, device="cuda:0", share: Optional["Program"] = None):
        """`share`: another loaded program; buffers of the same name (constants exported by the same process carry the address they
        had there: the weights, the prompt's K / V^T) are bound to ITS memory instead of a second copy — how a prompt-binding program
        (export_bind_prompt) writes the constants a denoise-step program reads."""
        self.device = torch.device(device)
        with open(path, "rb") as f:
            head = f.read(8 + 16 + 16)
            if head[:8] != MAGIC:
                raise ProgramError(f"{path} is not a step program")
            head_len, _total = struct.unpack("<qq", head[24:40])
            f.seek(0)
            blob = f.read(head_len)
            lib = hip.load()
            lib.mf_program_num_buffers.restype = C.c_int32
            lib.mf_program_find_buffer.restype = C.c_int32
            self._h = C.c_void_p()
            hip._check(lib.mf_program_load(blob, C.c_int64(len(blob)), C.byref(self._h)), "mf_program_load")
            self.tensors: Dict[int, torch.Tensor] = {}
            self.names: Dict[str, int] = {}
            with torch.cuda.device(self.device):
                for i in range(lib.mf_program_num_buffers(self._h)):
                    kind, nbytes, off, name = C.c_int32(), C.c_int64(), C.c_int64(), C.c_char_p()
                    hip._check(lib.mf_program_buffer_info(self._h, i, C.byref(kind), C.byref(nbytes), C.byref(off), C.byref(name)), "mf_program_buffer_info")
                    nm = name.value.decode()
                    if share is not None and kind.value != KIND_WORKSPACE and nm in share.names:
                        t = share.tensors[share.names[nm]]
                        if t.numel() < nbytes.value:
                            raise ProgramError(f"shared buffer {nm!r}: {t.numel()} bytes there, {nbytes.value} here")
                        self.tensors[i], self.names[nm] = t, i
                        hip._check(lib.mf_program_bind(self._h, i, C.c_void_p(t.data_ptr())), "mf_program_bind")
                        continue
                    t = torch.empty(max(nbytes.value, 1), dtype=torch.uint8, device=self.device)
                    if off.value >= 0:
                        f.seek(off.value)
                        t.copy_(torch.frombuffer(bytearray(f.read(nbytes.value)), dtype=torch.uint8))
                    self.tensors[i] = t
                    self.names[name.value.decode()] = i
                    hip._check(lib.mf_program_bind(self._h, i, C.c_void_p(t.data_ptr())), "mf_program_bind")

    def buffer(self, name: str, dtype=torch.uint8) -> torch.Tensor:
        return self.tensors[self.names[name]].view(dtype)

    @property
    def meta(self) -> str:
        lib = hip.load()
        lib.mf_program_meta.restype = C.c_char_p
        return lib.mf_program_meta(self._h).decode()

    @property
    def num_calls(self) -> int:
        return int(hip.load().mf_program_num_calls(self._h))

    def run(self) -> None:
        hip._check(hip.load().mf_program_run(self._h, hip._stream()), "mf_program_run")

    def close(self) -> None:
        if self._h:
            hip.load().mf_program_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _residual_list(down, mid, up) -> list:
    return list(down) + [mid] + list(up)


def export_brushnet(model, path: str, sample: torch.Tensor, temb: torch.Tensor, cond: torch.Tensor, conditioning_scale=1.0,
                    added_cond_kwargs=None) -> dict:
    """BrushNetModel.forward (models/brushnet.py:693-936) as a program for mf_brushnet_forward: io buffers "sample" (NCHW fp32
    latents), "temb" (a row block of model.time_embedding_table), "cond" (the conditioning latents) and "residual.<i>" — the 28
    residuals in the reference's order (down, mid, up), in the layout `layouts` of the returned dict reports (NHWC memory)."""
    import json
    t = torch.zeros(1, device=sample.device)

    def run():
        return model(sample, t, encoder_hidden_states=None, brushnet_cond=cond, conditioning_scale=conditioning_scale,
                     added_cond_kwargs=added_cond_kwargs, return_dict=False, _temb=temb)
    run()                                     # warm: tiles tuned, scratch sized
    with Recorder(dict(sample=sample, temb=temb, cond=cond)) as rec:
        res = _residual_list(*run())
        for i, r in enumerate(res):
            rec.output(f"residual.{i}", r)
    meta = dict(entry="mf_brushnet_forward", reference="models/brushnet.py:693-936", precision=model.prec.name, residuals=len(res), layouts=rec.layouts)
    info = rec.save(path, meta=json.dumps(meta))
    info["meta"] = meta
    return info


def export_unet(model, path: str, sample: torch.Tensor, temb: torch.Tensor, encoder_hidden_states: torch.Tensor, down, mid, up,
                added_cond_kwargs=None) -> dict:
    """UNet2DConditionModel.forward with BrushNet's residuals (models/unets/unet_2d_condition.py:1037-1311) as a program for
    mf_unet_forward: io buffers "sample", "temb", "residual.<i>" (the tensors passed here, any dense layout BrushNet returns) and
    "eps" (NCHW fp32).  The prompt is a constant of the program (its cross-attention K / V^T)."""
    import json
    res = _residual_list(down, mid, up)
    t = torch.zeros(1, device=sample.device)

    def run():
        return model(sample, t, encoder_hidden_states=encoder_hidden_states, down_block_add_samples=list(down), mid_block_add_sample=mid,
                     up_block_add_samples=list(up), added_cond_kwargs=added_cond_kwargs, return_dict=False, _temb=temb)[0]
    run()
    named = dict(sample=sample, temb=temb)
    for i, r in enumerate(res):
        named[f"residual.{i}"] = r
    with Recorder(named) as rec:
        eps = run()
        rec.output("eps", eps)
    meta = dict(entry="mf_unet_forward", reference="models/unets/unet_2d_condition.py:1037-1311", precision=model.prec.name, residuals=len(res),
                layouts=rec.layouts)
    info = rec.save(path, meta=json.dumps(meta))
    info["meta"] = meta
    return info


def export_vae_decode(vae, path: str, z: torch.Tensor) -> dict:
    """AutoencoderKL.decode (autoencoder_kl.py:294-318) as a program for mf_vae_decode: io buffers "z" (NCHW fp32, latents already divided by
    the scaling factor as pipeline_brushnet.py:1342 does) and "image" (NCHW fp32)."""
    import json
    z = z.to(vae.device).float().contiguous()
    vae.decode(z, return_dict=False)
    with Recorder(dict(z=z)) as rec:
        img = vae.decode(z, return_dict=False)[0]
        rec.output("image", img)
    meta = dict(entry="mf_vae_decode", reference="models/autoencoders/autoencoder_kl.py:294-318", precision=vae.prec.name, layouts=rec.layouts)
    info = rec.save(path, meta=json.dumps(meta))
    info["meta"] = meta
    return info


def export_vae_encode(vae, path: str, image: torch.Tensor) -> dict:
    """AutoencoderKL.encode up to the moments (autoencoder_kl.py:256-291) as a program for mf_vae_encode_moments: io buffers "image" (NCHW
    fp32) and "moments" (NHWC fp32 mean | logvar; mf_vae_sample draws the posterior sample from it, pipeline_brushnet.py:1188)."""
    import json
    image = image.to(vae.device).float().contiguous()
    vae._moments(image)
    with Recorder(dict(image=image)) as rec:
        m = vae._moments(image)
        rec.output("moments", m)
    meta = dict(entry="mf_vae_encode_moments", reference="models/autoencoders/autoencoder_kl.py:256-291", precision=vae.prec.name, layouts=rec.layouts)
    info = rec.save(path, meta=json.dumps(meta))
    info["meta"] = meta
    return info


def export_bind_prompt(unet, path: str) -> dict:
    """What depends on the prompt alone — K and V^T of every cross-attention layer (attention_processor.py:1246-1252: to_k / to_v of
    the prompt embeddings) — as a program: io buffer "prompt_embeds" ([2B, 77, C] in the model's storage dtype, [negative | positive]
    under classifier-free guidance, pipeline_brushnet.py:1103); the K / V^T it writes are the constants a denoise-step program exported
    by the same process reads (same buffer names: bind both to the same memory, `Program(..., share=step)`), so a host changes the
    prompt without exporting the step again.  Call after the UNet has run once with a prompt of the final shape."""
    import json
    from . import ops
    if not unet._cross_kv or getattr(unet, "_ehs_val", None) is None:
        raise ProgramError("export_bind_prompt: run the UNet (or the pipeline) once first: the K / V^T buffers do not exist yet")
    ctx = unet._ehs_val
    skv = ctx.shape[1]

    def run():
        for b, kv in list(unet._cross_kv.items()):
            ops.linear(ctx, unet.P[b + "to_k"], out=kv[0])
            ops.linear_t(ctx, unet.P[b + "to_v"], (skv + 7) // 8 * 8, out=kv[1])
    run()
    with Recorder(dict(prompt_embeds=ctx)) as rec:
        run()
    meta = dict(entry="mf_program_run", what="cross-attention K / V^T of the prompt", reference="models/attention_processor.py:1246-1252",
                precision=unet.prec.name, layers=len(unet._cross_kv), layouts=rec.layouts)
    info = rec.save(path, meta=json.dumps(meta))
    info["meta"] = meta
    return info
