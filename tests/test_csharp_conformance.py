"""Static checks of the C# binding sources, which no toolchain in the build container can compile:

  * csharp/GymxNative.cs against include/gymx.h -- every [LibraryImport] name and argument count, the field order and
    sizes of the mirrored structs, the enum values;
  * csharp/ReplayRecorder.cs against the reference's sources -- every private field it reads by reflection exists in
    the class it is read from, with the type it is read as (skipped where the reference checkout is absent);
  * csharp/GpuRegistration.cs -- GpuEnv.Seed re-keys the counter RNG the way gym_amd/envs.py's DeviceEnv.Seed does.

A typo in one of these would otherwise only show on a machine with a .NET SDK."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference/Gymnasium"


def _read(*parts):
    with open(os.path.join(ROOT, *parts)) as f:
        return f.read()


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


HEADER = _strip_comments(_read("include", "gymx.h"))
NATIVE = _strip_comments(_read("csharp", "GymxNative.cs"))


def _norm(name):
    return name.replace("_", "").lower()


# ------------------------------------------------------------------ functions ---
def _split_args(arglist):
    arglist = arglist.strip()
    if arglist in ("", "void"):
        return []
    return [a.strip() for a in arglist.split(",")]


def header_functions():
    out = {}
    for m in re.finditer(r"^(?:int|uint32_t|const char\*)\s+(gx_\w+)\s*\(([^;]*?)\)\s*;", HEADER, re.M | re.S):
        out[m.group(1)] = _split_args(" ".join(m.group(2).split()))
    return out


def csharp_imports():
    out = {}
    for m in re.finditer(r"\[LibraryImport\(Lib\)\]\s*internal static partial\s+(\w+)\s+(gx_\w+)\s*\(([^;]*?)\)\s*;", NATIVE, re.S):
        out[m.group(2)] = (m.group(1), _split_args(m.group(3)))
    return out


def test_every_entry_point_is_imported_with_the_declared_argument_count():
    h, cs = header_functions(), csharp_imports()
    assert len(h) >= 30
    assert set(h) == set(cs), sorted(set(h) ^ set(cs))
    for name, args in h.items():
        ret, cargs = cs[name]
        assert len(args) == len(cargs), (name, args, cargs)
    assert cs["gx_version"][0] == "uint" and cs["gx_last_error"][0] == "IntPtr"
    assert all(ret == "int" for name, (ret, _) in cs.items() if name not in ("gx_version", "gx_last_error"))


def test_argument_kinds_agree():
    """pointer / 64-bit / 32-bit per argument: a C `int` must not meet a C# IntPtr or the reverse"""
    h, cs = header_functions(), csharp_imports()

    def ckind(a):
        if "*" in a:
            return "ptr"
        t = a.split()[0] if not a.startswith("const") else a.split()[1]
        return {"int": "i32", "uint64_t": "i64", "size_t": "i64", "uint32_t": "i32"}[t]

    def cskind(a):
        a = re.sub(r"^(out|ref|in)\s+", "ptr ", a)
        if a.startswith("ptr ") or "*" in a:
            return "ptr"
        t = a.split()[0]
        return {"int": "i32", "uint": "i32", "ulong": "i64", "nuint": "i64", "IntPtr": "ptr"}[t]

    for name, args in h.items():
        got = [cskind(a) for a in cs[name][1]]
        want = [ckind(a) for a in args]
        assert got == want, (name, want, got)


# -------------------------------------------------------------------- structs ---
C_SIZES = {"uint32_t": 4, "int32_t": 4, "float": 4, "uint64_t": 8, "int64_t": 8, "double": 8}
CS_SIZES = {"uint": 4, "int": 4, "float": 4, "ulong": 8, "long": 8, "double": 8, "IntPtr": 8}


def header_struct(name):
    m = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), HEADER, re.S)
    assert m, name
    fields = []
    for decl in m.group(1).split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        if "*" in decl:
            fields.append((decl.replace("*", " ").split()[-1], 8))
            continue
        ctype, names = decl.split(" ", 1)
        for n in names.split(","):
            fields.append((n.strip(), C_SIZES[ctype]))
    return fields


def csharp_struct(name):
    m = re.search(r"public (?:unsafe )?struct %s\s*\{(.*?)\}" % name, NATIVE, re.S)
    assert m, name
    fields = []
    for decl in m.group(1).split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        assert decl.startswith("public "), decl
        decl = decl[len("public "):]
        ctype, names = decl.split(" ", 1)
        size = 8 if ctype.endswith("*") else CS_SIZES[ctype]
        for n in names.split(","):
            fields.append((n.strip(), size))
    return fields


def _layout(fields):
    """(offsets, total size) under natural alignment, which is both the C ABI's and LayoutKind.Sequential's"""
    off, offs = 0, []
    for _, size in fields:
        off = (off + size - 1) // size * size
        offs.append(off)
        off += size
    align = max(size for _, size in fields)
    return offs, (off + align - 1) // align * align


@pytest.mark.parametrize("c_name,cs_name", [("gx_desc", "GxDesc"), ("gx_info", "GxInfo"), ("gx_view", "GxView"), ("gx_stats", "GxStats"),
                                            ("gx_rows", "GxRows"), ("gx_transition", "GxTransition")])
def test_struct_fields_match_in_order_name_and_size(c_name, cs_name):
    c, cs = header_struct(c_name), csharp_struct(cs_name)
    assert [_norm(n) for n, _ in c] == [_norm(n) for n, _ in cs], (c, cs)
    assert [s for _, s in c] == [s for _, s in cs]
    assert _layout(c) == _layout(cs)


def test_struct_layouts_equal_the_ctypes_mirror():
    import ctypes as C

    from gym_amd import _lib as L

    for c_name, ct in [("gx_desc", L.GxDesc), ("gx_info", L.GxInfo), ("gx_view", L.GxView), ("gx_stats", L.GxStats),
                       ("gx_rows", L.GxRows), ("gx_transition", L.GxTransition)]:
        fields = header_struct(c_name)
        offs, total = _layout(fields)
        assert total == C.sizeof(ct), c_name
        assert [getattr(ct, f[0]).offset for f in ct._fields_] == offs, c_name
        assert [_norm(f[0]) for f in ct._fields_] == [_norm(n) for n, _ in fields], c_name


# ---------------------------------------------------------------------- enums ---
def header_enum(name):
    m = re.search(r"typedef enum %s \{(.*?)\} %s;" % (name, name), HEADER, re.S)
    assert m, name
    out = {}
    for item in m.group(1).split(","):
        item = item.strip()
        if item:
            k, v = item.split("=")
            out[k.strip()] = int(v.strip().rstrip("u"), 0)
    return out


def csharp_enum(name):
    m = re.search(r"public enum %s(?:\s*:\s*\w+)?\s*\{(.*?)\}" % name, NATIVE, re.S)
    assert m, name
    out = {}
    for item in m.group(1).split(","):
        item = item.strip()
        if item:
            k, v = item.split("=")
            out[k.strip()] = int(v.strip(), 0)
    return out


ALIASES = {"mountaincarcont": "mountaincarcontinuous", "acrobotrk4": "acrobotrk4"}


@pytest.mark.parametrize("c_name,cs_name,prefix,skip", [
    ("gx_status", "GxStatus", "GX_ERR_", {"GX_OK"}), ("gx_env_id", "GxEnvId", "GX_ENV_", {"GX_ENV_COUNT"}),
    ("gx_mode", "GxMode", "GX_MODE_", set()), ("gx_flags", "GxFlags", "GX_", set()), ("gx_policy", "GxPolicy", "GX_POLICY_", set()),
    ("gx_rollout_out", "GxRolloutOut", "GX_OUT_", set()), ("gx_dtype", "GxDType", "GX_", set())])
def test_enum_values_match(c_name, cs_name, prefix, skip):
    c = {k: v for k, v in header_enum(c_name).items() if k not in skip}
    cs = {k: v for k, v in csharp_enum(cs_name).items() if k not in ("None", "Ok") or c_name == "gx_rollout_out"}
    by_name = {}
    for k, v in c.items():
        n = _norm(k[len(prefix):] if k.startswith(prefix) else k)
        by_name[ALIASES.get(n, n)] = v
    assert by_name == {_norm(k): v for k, v in cs.items()}, (by_name, cs)


# ------------------------------------------------------------- the recorder ---
RECORDER = _strip_comments(_read("csharp", "ReplayRecorder.cs"))


def recorder_reads():
    """[(registry id, reference class, field name, C# type read as)]"""
    helpers = {}
    for m in re.finditer(r"private static void (\w+)\(object env, double\[\] r\)\s*\{(.*?)\}\s*$", RECORDER, re.M):
        helpers[m.group(1)] = re.findall(r"Field<([^>]*(?:\([^)]*\))?[^>]*)>\(env, \"(\w+)\"\)", m.group(2))
    out = []
    body = RECORDER[RECORDER.index("Layouts = new()"):]
    entries = re.split(r"\[\"([\w-]+)\"\]\s*=\s*new\(", body)[1:]
    for env_id, text in zip(entries[0::2], entries[1::2]):
        cls = re.search(r"\(\)\s*=>\s*new (\w+)\(\)", text).group(1)
        for t, f in re.findall(r"Field<((?:\([^)]*\)|[^>])+)>\(e, \"(\w+)\"\)", text):
            out.append((env_id, cls, f, t.replace(" ", "")))
        for h, reads in helpers.items():
            if re.search(r"\b%s\(e, r\)" % h, text):
                out.extend((env_id, cls, f, t.replace(" ", "")) for t, f in reads)
    return out


def reference_fields(cls):
    """private fields of the reference class: name -> type (spaces removed)"""
    src = _strip_comments(open(os.path.join(REFERENCE, "Envs.%s.cs" % cls)).read())
    fields = {}
    for m in re.finditer(r"^\s*private\s+(?:readonly\s+)?((?:\([^)]*\)|[\w<>\[\],]+?))\s+(_\w+(?:\s*,\s*_\w+)*)\s*(?:=[^;]*)?;", src, re.M):
        for name in m.group(2).split(","):
            fields[name.strip()] = m.group(1).replace(" ", "")
    return fields


def test_recorder_layouts_cover_what_they_claim():
    reads = recorder_reads()
    ids = {r[0] for r in reads}
    assert ids == {"CartPole-v1", "MountainCar-v0", "MountainCarContinuous-v0", "Pendulum-v1", "Acrobot-v1", "FrozenLake-v1",
                   "Taxi-v3", "CliffWalking-v0", "Blackjack-v1", "Pong-v4", "Breakout-v4"}
    assert len(reads) == 37   # 9 envs x (state, steps) + Pong 9 + Breakout 10
    # every layout reads the step counter, and the header's field counts equal the reads that fill them
    for env_id in ids:
        assert any(f == "_steps" for i, _, f, _ in reads if i == env_id), env_id


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference checkout is not on this machine")
def test_every_field_the_recorder_reflects_on_exists_with_that_type():
    for env_id, cls, field, read_as in recorder_reads():
        fields = reference_fields(cls)
        assert field in fields, "%s: Envs.%s.cs has no private field %s (has %s)" % (env_id, cls, field, sorted(fields))
        assert fields[field] == read_as, "%s: %s.%s is %s in the reference, the recorder reads it as %s" % (env_id, cls, field, fields[field], read_as)


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference checkout is not on this machine")
def test_recorder_constructs_classes_the_reference_registers_under_those_ids():
    reg = open(os.path.join(REFERENCE, "GymnasiumRegistration.cs")).read()
    for env_id, cls in {(r[0], r[1]) for r in recorder_reads()}:
        assert re.search(r'Register\("%s",\s*\(\)\s*=>\s*new (?:Envs\.)?%s\(\)\)' % (re.escape(env_id), cls), reg), (env_id, cls)


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference checkout is not on this machine")
def test_shells_declare_the_spaces_of_the_classes_they_stand_in_for():
    """ActionSpace sizes of the C# shells equal the reference's `new Discrete(n)` (agents read them through dynamic)"""
    shells = _strip_comments(_read("csharp", "GpuRegistration.cs"))
    for cls, ref in [("GpuCartPole", "CartPole"), ("GpuMountainCar", "MountainCar"), ("GpuAcrobot", "Acrobot"), ("GpuFrozenLake", "FrozenLake"),
                     ("GpuTaxi", "Taxi"), ("GpuCliffWalking", "CliffWalking"), ("GpuBlackjack", "Blackjack")]:
        body = shells[shells.index("class %s " % cls):]
        got = int(re.search(r"Discrete ActionSpace \{ get; \} = new\((\d+)\)", body).group(1))
        src = open(os.path.join(REFERENCE, "Envs.%s.cs" % ref)).read()
        want = int(re.search(r"ActionSpace\s*(?:\{ get; \})?\s*=>?\s*new\s*(?:Discrete)?\((\d+)\)", src).group(1))
        assert got == want, (cls, got, want)


# ----------------------------------------------------------------------- Seed ---
def test_csharp_seed_rekeys_like_the_python_mirror():
    cs = _strip_comments(_read("csharp", "GpuRegistration.cs"))
    m = re.search(r"public override void Seed\(int\? seed = null\)\s*\{(.*?)\n    \}", cs, re.S)
    assert m, "GpuEnv does not override Env.Seed(int? seed = null) (Env.cs:33-36)"
    body = " ".join(m.group(1).split())
    py = _read("gym_amd", "envs.py")
    pm = re.search(r"def Seed\(self, seed=None\):(.*?)\n    def ", py, re.S)
    pbody = " ".join(pm.group(1).split())
    # both: nothing on null; the seed becomes the key, the stream index goes back to 0, the batch is dropped and rebuilt lazily
    assert "if (seed is null) return;" in body and "if seed is not None:" in pbody
    assert "_seed = unchecked((ulong)(long)seed.Value);" in body and "self._seed = int(seed)" in pbody
    assert "_stream = 0;" in body and "self._stream = 0" in pbody
    assert "_vec?.Dispose(); _vec = null;" in body and "self._batch.close()" in pbody and "self._batch = None" in pbody
    # and the batch is built from exactly these two values
    assert "_vec ??= _make(_seed, _stream)" in cs
    assert "seed=self._seed" in py and "env_index_base=self._stream" in py
    assert len(re.findall(r"base\(\(seed, stream\) => new\(", cs)) == len(re.findall(r"seed: seed, envIndexBase: stream", cs)) >= 10
