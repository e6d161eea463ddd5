"""Step programs without a GPU: the recorder's signature table, the replay thunks' table in csrc/program.hip and the prototypes of
include/mfhip.h must say the same thing about every replayable entry (a wrong letter would pass a float where an int64 is read)."""
import os
import re

from reflecting_reality_amd import program

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_signatures():
    text = open(os.path.join(ROOT, "include", "mfhip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\bint\s+(mf_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        name, args = m.group(1), [a.strip() for a in m.group(2).replace("\n", " ").split(",")]
        if not args or "stream" not in args[-1]:
            continue
        sig = ""
        for a in args[:-1]:
            if "mf_gemm_desc" in a or "mf_groupnorm_desc" in a and "bwd" not in a:
                sig += "d"
            elif "*" in a:
                sig += "p"
            elif a.startswith("int64_t"):
                sig += "l"
            elif a.startswith("int32_t") or a.startswith("int "):
                sig += "i"
            elif a.startswith("float"):
                sig += "f"
            else:
                sig += "?"
        out[name] = sig
    return out


def test_signature_tables_agree_with_the_header():
    hdr = _header_signatures()
    for name, sig in program.SIGNATURES.items():
        assert name in hdr, f"{name} is not declared in include/mfhip.h with a trailing stream argument"
        assert hdr[name] == sig, f"{name}: program.SIGNATURES says {sig!r}, the header's prototype reads {hdr[name]!r}"
    src = open(os.path.join(ROOT, "reflecting-reality_amd", "csrc", "program.hip")).read()
    table = dict(re.findall(r'\{"(mf_\w+)",\s*"(\w+)"\}', src))
    assert table == program.SIGNATURES, "csrc/program.hip's kFns and program.SIGNATURES differ"
    # one thunk per table entry, and every thunk reads exactly the arguments its signature has
    enum = re.search(r"enum Fn \{(.*?)F_COUNT", src, flags=re.S).group(1)
    assert len([e for e in enum.replace("\n", " ").split(",") if e.strip()]) == len(table)


def test_two_d_copy_decomposition():
    import torch
    a = torch.zeros(4, 6, 8)
    assert program._rows_runs(a) == (1, 192, 192)
    assert program._rows_runs(a[:, :, :4]) == (24, 8, 4)
    assert program._rows_runs(a[:, :3]) == (4, 48, 24)
    assert program._rows_runs(a[:, ::2, :4]) == (12, 16, 4)             # every second row: still one pitch
    try:
        program._rows_runs(a[:, :2, :4])                               # rows 8 apart, then a jump of 48: two pitches
    except program.ProgramError:
        pass
    else:
        raise AssertionError("a copy with two different outer pitches is not a 2-D copy")


def _blob(calls, buffers, nevents=0, meta="{}"):
    return program.serialize_header(calls, buffers, nevents, meta)[0]


def test_program_header_round_trips_through_the_library_without_a_gpu():
    """program.serialize_header -> mf_program_load: the parser accepts what the writer writes (buffers, names, data offsets, meta, calls on
    two streams with an event between them) and refuses malformed or foreign input with a message — no device call is made."""
    import ctypes as C
    import struct
    from reflecting_reality_amd import hip
    lib = hip.load()
    lib.mf_program_num_buffers.restype = C.c_int32
    lib.mf_program_num_calls.restype = C.c_int32
    lib.mf_program_find_buffer.restype = C.c_int32
    lib.mf_program_meta.restype = C.c_char_p
    bufs = [dict(kind=program.KIND_IO, name="x", bytes=256), dict(kind=program.KIND_WORKSPACE, name="workspace.0", bytes=4096),
            dict(kind=program.KIND_CONST, name="const.abc", bytes=512)]
    calls = [("@record", [(program.A_I32, 0)], 0), ("@wait", [(program.A_I32, 0)], 1),
             ("mf_silu_f32", [(program.A_PTR, 0, 0), (program.A_PTR, 1, 128), (program.A_I64, 64)], 1),
             ("mf_cfg_combine", [(program.A_PTR, 0, 0), (program.A_PTR, -1, 0), (program.A_F32, 7.5), (program.A_PTR, 2, 16), (program.A_I64, 8)], 0)]
    blob = _blob(calls, bufs, nevents=1, meta='{"entry": "test"}')
    h = C.c_void_p()
    assert lib.mf_program_load(blob, C.c_int64(len(blob)), C.byref(h)) == 0, lib.mf_last_error()
    assert lib.mf_program_num_buffers(h) == 3 and lib.mf_program_num_calls(h) == 4 and lib.mf_program_meta(h) == b'{"entry": "test"}'
    assert lib.mf_program_find_buffer(h, b"const.abc") == 2 and lib.mf_program_find_buffer(h, b"nope") == -1
    kind, nbytes, off, name = C.c_int32(), C.c_int64(), C.c_int64(), C.c_char_p()
    assert lib.mf_program_buffer_info(h, 1, C.byref(kind), C.byref(nbytes), C.byref(off), C.byref(name)) == 0
    assert (kind.value, nbytes.value, off.value, name.value) == (program.KIND_WORKSPACE, 4096, -1, b"workspace.0")
    assert lib.mf_program_buffer_info(h, 2, C.byref(kind), C.byref(nbytes), C.byref(off), C.byref(name)) == 0
    assert off.value % 256 == 0 and off.value >= len(blob) + 256 and nbytes.value == 512
    assert lib.mf_program_bind(h, 0, C.c_void_p(0x1008)) != 0 and b"aligned" in lib.mf_last_error()
    assert lib.mf_program_bind(h, 7, C.c_void_p(0x1000)) != 0
    assert lib.mf_program_bind(h, 0, C.c_void_p(0x1000)) == 0
    assert lib.mf_program_run(h, None) != 0 and b"not bound" in lib.mf_last_error()       # refused before anything is launched
    lib.mf_program_destroy(h)
    # refusals: an entry without a thunk, a signature mismatch, a buffer index out of range, a stream the header does not declare, a foreign ABI
    for bad_calls, nev, what in (([("mf_adamw", [(program.A_PTR, 0, 0)], 0)], 0, b"no replay thunk"),
                                 ([("mf_silu_f32", [(program.A_PTR, 0, 0), (program.A_I64, 1), (program.A_I64, 64)], 0)], 0, b"signature"),
                                 ([("mf_silu_f32", [(program.A_PTR, 0, 0), (program.A_PTR, 9, 0), (program.A_I64, 64)], 0)], 0, b"signature"),
                                 ([("@wait", [(program.A_I32, 3)], 0)], 1, b"malformed")):
        b2 = _blob(bad_calls, bufs, nevents=nev)
        assert lib.mf_program_load(b2, C.c_int64(len(b2)), C.byref(h)) != 0 and what in lib.mf_last_error(), (what, lib.mf_last_error())
    b3 = bytearray(blob)
    b3[8:12] = struct.pack("<I", hip.ABI_VERSION + 1)
    assert lib.mf_program_load(bytes(b3), C.c_int64(len(b3)), C.byref(h)) != 0 and b"ABI" in lib.mf_last_error()
    assert lib.mf_program_load(blob[:-8], C.c_int64(len(blob) - 8), C.byref(h)) != 0
