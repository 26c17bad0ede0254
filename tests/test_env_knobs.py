"""DESIGN.md section 8 lists every environment variable the library reads: this test greps getenv("...") in
opengpc_amd/csrc and include/ and holds the table against it (both ways), so a knob cannot be added, renamed or removed
without the document following."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def variables_read_by_the_library():
    found = {}
    for base in (os.path.join(ROOT, "opengpc_amd", "csrc"), os.path.join(ROOT, "include")):
        for dirpath, _, files in os.walk(base):
            for f in files:
                if not f.endswith((".h", ".hpp", ".hip", ".cpp")):
                    continue
                text = open(os.path.join(dirpath, f), errors="replace").read()
                for name in re.findall(r'getenv\(\s*"([A-Za-z_0-9]+)"\s*\)', text):
                    found.setdefault(name, []).append(os.path.relpath(os.path.join(dirpath, f), ROOT))
    return found


def variables_in_the_design_table():
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    sec = text[text.index("## 8. Environment variables"):text.index("## 9.")]
    names = set()
    for line in sec.splitlines():
        if not line.startswith("| `"):
            continue
        first = line.split("|")[1]
        names.update(re.findall(r"`([A-Z][A-Z_0-9]+)`", first))
    return names


def test_every_variable_the_library_reads_is_in_the_table_and_nothing_else_is():
    read = variables_read_by_the_library()
    listed = variables_in_the_design_table()
    assert len(read) >= 25
    missing = sorted(set(read) - listed)
    stale = sorted(listed - set(read))
    assert not missing, "read by the library but not in DESIGN.md section 8: %s (%s)" % (missing, [read[m] for m in missing])
    assert not stale, "in DESIGN.md section 8 but no getenv reads them: %s" % stale


def test_knobs_are_read_at_context_creation_not_per_launch():
    """(round-4 advice) no getenv on a launch path: inside gpc_hip.hip every getenv sits in gpc_hip_create or in one of
    the once-per-context helpers it calls."""
    text = open(os.path.join(ROOT, "opengpc_amd", "csrc", "gpc_hip.hip")).read()
    allowed = ("int gpc_hip_create(", "void find_gpu_node_cpus(", "int default_expand_threads(")
    starts = sorted((m.start(), m.group(0)) for m in re.finditer(r"^(?:static |inline )?(?:[\w:<>*&]+ )+\w+\([^;{]*\)\s*(?:const\s*)?\{", text, re.M))
    for m in re.finditer(r'getenv\("', text):
        owner = [s for s in starts if s[0] < m.start()][-1][1]
        assert any(a in owner.replace("\n", " ") for a in allowed), owner
