"""The data_custom loader of the host side (host/bal_problem.cpp, restating bal_problem.cpp:183-303): whole file in
memory, tokens parsed in place on several threads.  It must read what `operator>>` / fscanf read: any whitespace and line
structure, every spelling of a number -- and give the correctly rounded double (compared with Python's float(), which is
correctly rounded too)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "loader_check")


def _load(path):
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "../../build/loader_check"],
                              stdout=subprocess.DEVNULL)
    r = subprocess.run([BIN, path], capture_output=True, text=True)
    return r.returncode, r.stdout, r.stderr


LONG = "0." + "123456789" * 9                      # 83 characters: longer than any stack copy of a token
SPELLINGS = ["0", "-0", "7", LONG, "-12.5", "+3.25", ".5", "-.125", "3.", "1e-3", "1E+3", "-2.5e-07", "123456.789012",
             "0.000001", "999999999999999", "1234567890123456789", "0.1234567890123456789", "1e22", "1e23", "1e-22",
             "1e-23", "4.9e-324", "1.7976931348623157e308", "0x1.8p1", "2.2250738585072014e-308", "123456789.123456789e-5",
             "00012.50", "1e0005"]


def test_every_spelling_of_a_number_is_read_like_strtod(tmp_path):
    rng = np.random.default_rng(0)
    n_c, n_l = 3, len(SPELLINGS)
    toks = [str(n_c), str(n_l), str(2 * n_l)]
    want_obs = []
    for l, sp in enumerate(SPELLINGS):            # two observations per landmark, descending camera order in the file
        for c in (2, 0):
            toks += [str(c), str(l), sp, SPELLINGS[(l + 1 + c) % n_l]]
            want_obs.append((l, c, float.fromhex(sp) if sp.startswith("0x") else float(sp),
                             -(float.fromhex(s2) if (s2 := SPELLINGS[(l + 1 + c) % n_l]).startswith("0x") else float(s2))))
    cams = [SPELLINGS[(7 * i) % n_l] for i in range(15 * n_c)]
    lms = [SPELLINGS[(5 * i + 1) % n_l] for i in range(3 * n_l)]
    toks += cams + lms
    seps = [" ", "\n", "\t", "  \n ", "\r\n", " \t "]
    text = "".join(t + seps[rng.integers(len(seps))] for t in toks)
    f = tmp_path / "spellings.txt"
    f.write_text(text)
    rc, out, err = _load(str(f))
    assert rc == 0, err
    lines = out.strip().splitlines()
    got_obs = [ln.split() for ln in lines if ln.startswith("obs")]
    want_obs.sort(key=lambda t: (t[0], t[1]))     # the loader keeps a landmark's cameras ascending
    assert len(got_obs) == len(want_obs)
    for g, w in zip(got_obs, want_obs):
        assert (int(g[1]), int(g[2])) == (w[0], w[1])
        assert float(g[3]) == w[2] and float(g[4]) == w[3], (g, w)
    val = lambda s: float.fromhex(s) if s.startswith("0x") else float(s)
    got_cam = [float(x) for ln in lines if ln.startswith("cam") for x in ln.split()[1:]]
    assert got_cam == [val(s) for s in cams]
    got_lm = [float(x) for ln in lines if ln.startswith("lm") for x in ln.split()[1:]]
    assert got_lm == [val(s) for s in lms]


def test_loader_matches_the_python_mirror_on_a_generated_file(tmp_path):
    """Several MB of text (the parallel path: one piece per MB), against synth.read_data_custom."""
    from povar_amd import synth
    p = synth.make_problem(60, 30000, 150000, seed=11)
    f = str(tmp_path / "problem.txt")
    synth.write_data_custom(f, p)
    rc, out, err = _load(f)
    assert rc == 0, err
    q = synth.read_data_custom(f)
    obs = np.array([[float(x) for x in ln.split()[1:]] for ln in out.splitlines() if ln.startswith("obs")])
    assert obs.shape[0] == q.n_obs
    lm_of = np.repeat(np.arange(q.n_lms), np.diff(q.lm_off))
    assert np.array_equal(obs[:, 0], lm_of) and np.array_equal(obs[:, 1], q.cam_idx)
    assert np.array_equal(obs[:, 2:], np.asarray(q.obs).reshape(-1, 2))
    cams = np.array([[float(x) for x in ln.split()[1:13]] for ln in out.splitlines() if ln.startswith("cam")])
    assert np.array_equal(cams, np.asarray(q.cams).reshape(-1, 12))


@pytest.mark.parametrize("damage", ["truncated", "letters", "duplicate", "index", "fraction-as-index", "glued"])
def test_loader_rejects_damaged_files(tmp_path, damage):
    """FATAL like the reference (bal_problem.cpp:227, 297-300): a short file, a non-number, a repeated (camera, landmark)
    pair, an index out of range; and a number that is only PART of its token -- a fraction where an index belongs, two
    values glued together -- which would otherwise shift every later field of the piece without an error."""
    head = "2 2 3\n0 0 1.0 2.0\n1 0 3.0 4.0\n"
    third = {"duplicate": "1 0 5.0 6.0\n", "index": "2 1 5.0 6.0\n"}.get(damage, "0 1 5.0 6.0\n")
    body = head + third + "\n".join(["0.5"] * 30) + "\n" + "\n".join(["1.5"] * 6) + "\n"
    if damage == "truncated":
        body = body[: len(body) - 20]
    if damage == "letters":
        body = body.replace("3.0", "abc", 1)
    if damage == "fraction-as-index":
        body = body.replace("1 0 3.0", "1.5 0 3.0", 1)
    if damage == "glued":
        body = body.replace("3.0 4.0", "3.0x4.0 9", 1)
    f = tmp_path / "bad.txt"
    f.write_text(body)
    rc, out, err = _load(str(f))
    assert rc != 0 and "FATAL" in err
