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
