"""The drop-in boundary without a GPU: libyhair.so loads, exports every symbol include/yhair.h
declares, fails loudly when no device is present (no CPU fallback), and its host-side scene
reader reproduces the reference loader's semantics. No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "yhair.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(yh_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(yh):
    lib = yh.load()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"libyhair.so does not export {name}"
    # and the Python binding covers the same set
    assert sorted(yh.EXPORTS) == declared
    assert b"gfx950" in lib.yh_version()


def test_code_object_is_gfx950_only(built):
    so = os.path.join(ROOT, "yocto-hair_amd", "libyhair.so")
    blob = open(so, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_no_product_file_references_the_oracle():
    """The product path must not import / link / call anything under oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "yocto-hair_amd")):
        for f in files:
            if f.endswith((".cpp", ".h", ".hip", ".py")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"yh_oracle|libyh_ref|yo_scene|oracle_capi|oracle/", text) and "never imports the oracle" not in text:
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_create_fails_loudly_without_gpu(yh):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = yh.load()
    assert lib.yh_create(0) is None
    assert b"no CPU fallback" in lib.yh_last_error(None)
    with pytest.raises(yh.YhError):
        yh.Context(0)


def test_scene_loader_semantics(yh):
    from conftest import scene_path
    sf = yh.SceneFile(scene_path("hair-curls", scale=0.05))
    d = sf.desc.contents
    # objects in alphabetical order of their JSON keys (nlohmann json = std::map):
    # arealight1, arealight2, black, blonde, brown, red
    assert d.num_objects == 6 and d.num_shapes == 2 and d.num_environments == 1
    mats = [d.materials[d.objects[i].material] for i in range(6)]
    assert [round(m.eumelanin, 3) for m in mats] == [0, 0, 8, 0.3, 1.3, 0]
    assert mats[5].pheomelanin == 2 and mats[0].emission[0] == 20
    assert all(abs(m.beta_n - 0.9) < 1e-7 for m in mats[2:])
    # lookat objects use the inv_xz frame (x and z flipped), translation = eye
    f = np.array(d.objects[0].frame[:])
    assert np.allclose(f[9:], [-8, 10, 5])
    z = f[6:9]
    assert np.allclose(z, -(np.array([-8, 10, 5]) - [-6.5, 3.5, 0]) / np.linalg.norm(np.array([-8, 10, 5]) - [-6.5, 3.5, 0]), atol=1e-6)
    # the instanced hair shape is shared
    assert len({d.objects[i].shape for i in range(2, 6)}) == 1
    hair = d.shapes[d.objects[2].shape]
    assert hair.num_lines == 500 * 100 and hair.num_triangles == 0 and bool(hair.radius) and bool(hair.normals)
    light = d.shapes[d.objects[0].shape]
    assert light.num_triangles == 2 and [light.triangles[i] for i in range(6)] == [0, 1, 2, 3, 2, 1]
    env = d.environments[0]
    assert (env.tex_width, env.tex_height) == (2048, 1024) and bool(env.texels)
    # square film for aspect 1.0 (pt.cpp:2088-2092), focus = |eye - center|
    assert np.isclose(d.camera.film[0], 0.036) and np.isclose(d.camera.film[1], 0.036)
    assert np.isclose(d.camera.focus, 20.0)
    sf.close()


def test_scene_loader_errors(yh, tmp_path):
    with pytest.raises(yh.YhError, match="file not found"):
        yh.SceneFile(str(tmp_path / "nope.json"))
    p = tmp_path / "s.json"
    p.write_text('{"cameras": {"c": {}}, "objects": {"o": {"shape": "missing"}}}')
    with pytest.raises(yh.YhError, match="missing.ply: file not found"):
        yh.SceneFile(str(p))
    p.write_text('{"cameras": {"c": {}}, "objects": {"o": {"shape": "x", "material": "m"}}}')
    with pytest.raises(yh.YhError, match="missing material m"):
        yh.SceneFile(str(p))


def test_save_image_pfm_layout(yh, tmp_path):
    lib = yh.load()
    img = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)
    err = C.create_string_buffer(256)
    path = str(tmp_path / "x.pfm")
    assert lib.yh_save_image(path.encode(), 3, 2, yh.fptr(img), err, 256) == 0
    raw = open(path, "rb").read()
    assert raw.startswith(b"PF\n3 2\n-1\n")
    data = np.frombuffer(raw[len(b"PF\n3 2\n-1\n"):], np.float32).reshape(2, 3, 3)
    assert np.array_equal(data, img[..., :3])  # top row first, rgb only (yocto_image.cpp:1527-1556)
    assert lib.yh_save_image(str(tmp_path / "x.png").encode(), 3, 2, yh.fptr(img), err, 256) == yh.YH_E_IO


def _shape_boxes(shape):
    """primitive bounds as yh_upload_scene computes them (line_bounds / triangle_bounds, math.h:3037-3044)"""
    pos = np.ctypeslib.as_array(shape.positions, (shape.num_vertices, 3))
    if shape.num_lines > 0:
        idx = np.ctypeslib.as_array(shape.lines, (shape.num_lines, 2))
        rad = np.ctypeslib.as_array(shape.radius, (shape.num_vertices,)) if shape.radius else np.full(shape.num_vertices, 0.001, np.float32)
        p, r = pos[idx], rad[idx][..., None]
        return np.concatenate([(p - r).min(axis=1), (p + r).max(axis=1)], axis=1).astype(np.float32)
    idx = np.ctypeslib.as_array(shape.triangles, (shape.num_triangles, 3))
    p = pos[idx]
    return np.concatenate([p.min(axis=1), p.max(axis=1)], axis=1).astype(np.float32)


@pytest.mark.parametrize("name,kw", [("sphere-hairblock", dict(scale=0.5)), ("hair-curls", dict(scale=0.05))])
def test_host_bvh_is_the_reference_tree(yh, oracle, name, kw):
    """The product host builder must produce the reference's tree node for node
    (boxes, child index, count, split axis, breadth-first numbering) and the same leaf order; the
    oracle's tree is itself pinned to the reference through bit-identical hits and images."""
    from conftest import scene_path
    sf = yh.SceneFile(scene_path(name, **kw))
    d = sf.desc.contents
    osc = oracle.scene(sf.desc)
    lib = yh.load()
    for si in range(d.num_shapes):
        boxes = np.ascontiguousarray(_shape_boxes(d.shapes[si]))
        n = lib.yh_bvh_build(len(boxes), yh.fptr(boxes), None, None)
        nodes, prims = np.zeros((n, 8), np.float32), np.zeros(len(boxes), np.int32)
        assert lib.yh_bvh_build(len(boxes), yh.fptr(boxes), yh.fptr(nodes), yh.iptr(prims)) == n
        want = osc.bvh(si)
        assert nodes.shape == want.shape
        assert np.array_equal(nodes.view(np.uint32), want.view(np.uint32))
        wp = np.zeros(len(boxes), np.int32)
        oracle.lib.yo_scene_bvh(osc.h, si, None, yh.iptr(wp))
        assert np.array_equal(prims, wp)
        # and it is deterministic
        nodes2 = np.zeros_like(nodes)
        lib.yh_bvh_build(len(boxes), yh.fptr(boxes), yh.fptr(nodes2), None)
        assert np.array_equal(nodes.view(np.uint32), nodes2.view(np.uint32))
    osc.close(), sf.close()


def _binary_leaf_order(nodes, sign, tests):
    """Leaves of the reference's binary tree in the order intersect_shape_bvh visits them (pt.cpp:887-893: near side of
    the split axis first) for a ray whose direction has sign bits `sign` and that enters every node. `tests`: in place
    of "enters every node", a set of binary node ids whose box the ray misses."""
    meta = nodes[:, 7].view(np.int32)
    start = nodes[:, 6].view(np.int32)
    out, stack = [], [0]
    while stack:
        i = stack.pop()
        if i in tests:
            continue
        internal, axis, num = (meta[i] >> 16) & 1, (meta[i] >> 24) & 3, meta[i] & 0xFFFF
        if not internal:
            out.append((int(start[i]), int(num)))
        elif (sign >> axis) & 1:  # direction negative on the axis: the second child is nearer
            stack.append(int(start[i])), stack.append(int(start[i]) + 1)
        else:
            stack.append(int(start[i]) + 1), stack.append(int(start[i]))
    return out


def _wide_leaf_order(slots, width, sign, missed_leaves):
    """The same for the collapsed tree, with the device's rank arithmetic (csrc/dev_trace.h: the node steps of
    YH_MODE_QUAD / _OCT / _HEX): rank bit per collapsed level = side bit xor near bit of that level's split axis."""
    levels = {4: 2, 8: 3, 16: 4}[width]
    at = [0, 2, 6, 14]
    ref = slots[:, :, 6].view(np.uint32)
    axes = slots[:, :, 7].view(np.uint32)
    out, stack = [], [0]
    while stack:
        cur = stack.pop()
        if cur & 0xC0000000 == 0xC0000000:
            leaf = (int(cur & 0x07FFFFFF), int((cur >> 27) & 7))
            if leaf not in missed_leaves:
                out.append(leaf)
            continue
        ranked = []
        for o in range(width):
            r = int(ref[cur, o])
            if r == 0xFFFFFFFF:
                continue
            ax = int(axes[cur, o])
            rank = 0
            for lv in range(levels):
                path = o >> (levels - lv)              # the side bits above this level
                side = (o >> (levels - 1 - lv)) & 1
                near = (sign >> ((ax >> (at[lv] + 2 * path)) & 3)) & 1
                rank |= (side ^ near) << (levels - 1 - lv)
            ranked.append((rank, r))
        for _, r in sorted(ranked, reverse=True):       # pushed so that they pop in visiting order
            stack.append(r)
    return out


@pytest.mark.parametrize("width", [4, 8, 16])
def test_wide_nodes_keep_the_reference_visiting_order(yh, width):
    """The 4- / 8- / 16-wide collapses of the reference's tree (host/bvh_build.cpp), without a GPU: every leaf of the
    binary tree appears once, slot boxes are the binary nodes' boxes, and the children ranked by the device's formula
    are visited in the order the reference's binary traversal visits them, for all eight direction-sign triples."""
    rng = np.random.default_rng(11)
    lib = yh.load()
    for n in (1, 3, 4, 5, 37, 1000, 5003):
        c = rng.uniform(-1, 1, (n, 3)).astype(np.float32) * np.array([1, 0.3, 2], np.float32)
        boxes = np.ascontiguousarray(np.concatenate([c - 0.01, c + 0.01], axis=1).astype(np.float32))
        nb = lib.yh_bvh_build(n, yh.fptr(boxes), None, None)
        nodes = np.zeros((nb, 8), np.float32)
        lib.yh_bvh_build(n, yh.fptr(boxes), yh.fptr(nodes), None)
        nw = lib.yh_bvh_build_wide(n, yh.fptr(boxes), width, None)
        slots = np.zeros((nw, width, 8), np.float32)
        assert lib.yh_bvh_build_wide(n, yh.fptr(boxes), width, yh.fptr(slots)) == nw
        for sign in range(8):
            want = _binary_leaf_order(nodes, sign, set())
            assert _wide_leaf_order(slots, width, sign, set()) == want, (n, width, sign)
            assert sorted(want) == sorted(set(want)) and sum(k for _, k in want) == n  # every primitive once
        # slot boxes: a leaf slot carries the box of its binary leaf
        meta, start = nodes[:, 7].view(np.int32), nodes[:, 6].view(np.int32)
        leaf_box = {(int(start[i]), int(meta[i] & 0xFFFF)): nodes[i, :6] for i in range(nb) if not (meta[i] >> 16) & 1}
        ref = slots[:, :, 6].view(np.uint32)
        for w, o in zip(*np.nonzero((ref & 0xC0000000) == 0xC0000000)):
            if ref[w, o] == 0xFFFFFFFF:
                continue
            key = (int(ref[w, o] & 0x07FFFFFF), int((ref[w, o] >> 27) & 7))
            assert np.array_equal(slots[w, o, :6], leaf_box[key])


def test_leaf_group_merge_is_the_sequential_rule():
    """Leaf groups (csrc/dev_trace.h, YH_MODE_OCTP / YH_MODE_HEXP) test up to four leaves of up to four primitives in one
    step, every primitive against the ray as it was BEFORE the step, and keep the accepted primitive of minimum t, the
    later one on a tie. The reference meets the primitives one after the other, each against the ray shortened by the
    hits before it (pt.cpp:905-923: `ray.tmax = distance` after every hit; the primitive test rejects only t > tmax,
    math.h:3450). Same survivor, also with ties and with misses in between: checked here on random cases."""
    rng = np.random.default_rng(7)
    for _ in range(20000):
        n = int(rng.integers(1, 17))
        t = rng.integers(1, 6, n).astype(np.float32)  # few distinct values: many exact ties
        hits_if_reachable = rng.random(n) < 0.6      # the primitive test's verdict apart from its t > tmax reject
        tmax0 = np.float32(rng.integers(2, 7))
        best, tmax = -1, tmax0  # the reference: sequential, shrinking tmax
        for i in range(n):
            if hits_if_reachable[i] and not t[i] > tmax:
                best, tmax = i, t[i]
        cand = [i for i in range(n) if hits_if_reachable[i] and not t[i] > tmax0]  # the device: all against the old tmax ...
        merged = -1
        for i in cand:  # ... then minimum t, the later index on a tie
            if merged < 0 or t[i] < t[merged] or (t[i] == t[merged] and i > merged):
                merged = i
        assert merged == best


def test_launch_deadline_logic(tmp_path):
    """The bounded wait behind yh_trace_samples / yh_synchronize (yocto-hair_amd/host/deadline.h), compiled
    with mocked calls: BoundedCall (what the library uses: the blocking synchronise on a worker thread, the caller waiting with the
    deadline) — a call that returns is DONE with its value, one that NEVER returns is EXPIRED at the deadline, every later call
    at once, the destructor does not wait for it, and a satisfied wait costs microseconds; wait_until (the polling form) with a
    mocked event query and a fake clock; YHAIR_LAUNCH_TIMEOUT_S parsed with the default for anything that is not a positive number."""
    exe = str(tmp_path / "test_deadline")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "yocto-hair_amd", "host"),
                           os.path.join(ROOT, "tests", "cpp", "test_deadline.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "all checks passed" in out.stdout, out.stdout + out.stderr


def test_eight_cold_ranks_share_one_trial_record(tmp_path):
    """Eight cold ranks on one node, rehearsed on CPU: tests/cpp/test_trial_ranks.cpp compiles the REAL
    host/launch_plan.cpp with the device mocked and drives its trial bookkeeping for eight ranks (threads) against ONE trials_v2.txt.
    cold: concurrent O_APPEND writers, every rank ends with a complete record, no trial inside any rank's timed steps
    (launches_in_timed_steps == steps), tied candidates settle on one kernel on all ranks. Then this test damages the file the ways a node can
    — a torn line (a rank killed inside its write), a damaged line, a second record for one key whose winner differs (two runs that disagreed:
    the last line counts), a torn line without its newline at the end of the file — and a second process (warm) must run no trial at all,
    one launch per request, write nothing, and pick the kernels the surviving lines name."""
    exe, cache = str(tmp_path / "test_trial_ranks"), str(tmp_path / "cache")
    os.makedirs(cache)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-D__HIP_PLATFORM_AMD__", "-Wno-unused-function", "-I/opt/rocm/include",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "yocto-hair_amd", "host"),
                           os.path.join(ROOT, "tests", "cpp", "test_trial_ranks.cpp"), "-o", exe, "-ldl"])
    out = subprocess.run([exe, "cold", cache], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "all checks passed" in out.stdout, out.stdout + out.stderr
    path = os.path.join(cache, "trials_v2.txt")
    lines = open(path).read().splitlines()
    assert len(lines) == 8 * 42

    def line_of(rank):  # the record of shard `rank` of 8 of the 720 x 720 image of scene 0xC1
        hits = [l for l in lines if f"|{0xC1:016x}|720|720|{rank}|8|8 =" in l]
        assert len(hits) == 1
        return hits[0]

    def with_time(line, shape, factor):
        key, _, rest = line.partition(" =")
        pairs, _, tail = rest.partition(" ;")
        pairs = pairs.split()
        ms, n = pairs[shape].split(":")
        pairs[shape] = f"{float(ms) * factor:.9g}:{n}"
        return key + " = " + " ".join(pairs) + " ;" + tail

    with open(path, "a") as f:
        f.write(line_of(3)[: len(line_of(3)) * 6 // 10] + "\n")      # torn: a rank died inside its write — the earlier line of the key stands
        f.write(with_time(line_of(4), 8, -1.5) + "\n")               # damaged: a negative time is not a record
        f.write(with_time(line_of(5), 8, 1.3) + "\n")                # a later run measured the leaf-group form 30 % slower: the LAST line counts -> shape 6
        f.write(with_time(line_of(6), 8, 1.3).rpartition(" ;")[0])    # ... and the same for rank 6, but torn at the very end of the file (no tail, no newline): ignored
    out = subprocess.run([exe, "warm", cache, "5:6"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "all checks passed" in out.stdout, out.stdout + out.stderr


def test_codegen_check_knows_two_round_trips_when_it_sees_them():
    """tools/check_codegen.py also guards k_stream's step against the regression round 5 found in the assembly: a wait between the step's
    buffer loads that an earlier one of them has to satisfy = two dependent memory round trips per step (the first form of the cooperative
    leaves: -15 % instructions, 0 % time). The checker's rule on two hand-made instruction streams."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_codegen", os.path.join(ROOT, "tools", "check_codegen.py"))
    cc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cc)
    own = ["\tbuffer_load_dwordx4 v[%d:%d], v10, s[68:71], 0 offen offset:%d" % (16 + 4 * k, 19 + 4 * k, 16 * k) for k in range(8)]
    seg = ["\tbuffer_load_dwordx4 v[8:11], v12, s[68:71], 0 offen", "\tbuffer_load_dwordx4 v[12:15], v12, s[68:71], 0 offen offset:16"]
    good = own + ["\tds_write_b64 v9, v[88:89]", "\ts_waitcnt lgkmcnt(0)", "\ts_waitcnt vmcnt(8)"] + seg + ["\ts_waitcnt vmcnt(2)", "\tv_sub_f32_e32 v1, v16, v2"]
    assert cc.loads_in_one_round_trip(good) == (10, None)
    bad = seg + ["\ts_load_dwordx2 s[4:5], s[34:35], 0x440", "\ts_waitcnt vmcnt(0) lgkmcnt(0)"] + own + ["\ts_waitcnt vmcnt(0)"]
    n, line = cc.loads_in_one_round_trip(bad)
    assert n == 10 and line == "s_waitcnt vmcnt(0) lgkmcnt(0)"
    assert cc.loads_in_one_round_trip(["\tv_mov_b32_e32 v0, 0"]) == (0, None)


def test_traversal_loops_do_not_spill():
    """The traversal loop of the product kernels (plain k_trace in both launch shapes, plain k_stream) must not contain
    scratch instructions: a spill reload there stalls every step of every ray (0.75-0.8x on the dense configs), and
    whether the register allocator puts one there flips with unrelated edits of the shading code (profiles/r02).
    tools/check_codegen.py compiles the kernels to gfx950 assembly (no GPU needed) and looks."""
    import shutil
    import subprocess
    import sys
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_codegen.py"), "--strict"], capture_output=True, text=True)
    if r.returncode != 0:
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
        if "7.2" not in ver:  # a register-allocator outcome: a statement about the pinned toolchain (ROCm 7.2) only
            pytest.xfail("traversal-loop spill check failed under another toolchain than ROCm 7.2:\n" + r.stdout[-1500:])
    assert r.returncode == 0, r.stdout + r.stderr
