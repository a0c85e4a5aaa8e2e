"""Rating-file readers and factor files (host only).  The fixtures are written by the
test in the public formats of MovieLens / Netflix Prize; no dataset is shipped or fetched."""
import numpy as np
import pytest

ROWS = [(196, 242, 3.0), (186, 302, 3.5), (22, 377, 1.0), (196, 51, 5.0), (7, 242, 4.0), (186, 9000001, 2.0)]


def _expect(d):
    uid = sorted({a for a, _, _ in ROWS})
    iid = sorted({b for _, b, _ in ROWS})
    np.testing.assert_array_equal(d["user_ids"], uid)
    np.testing.assert_array_equal(d["item_ids"], iid)
    np.testing.assert_array_equal(d["u"], [uid.index(a) for a, _, _ in ROWS])
    np.testing.assert_array_equal(d["i"], [iid.index(b) for _, b, _ in ROWS])
    np.testing.assert_array_equal(d["r"], np.array([c for _, _, c in ROWS], np.float32))
    assert d["n_users"] == len(uid) and d["n_items"] == len(iid)


def test_movielens_100k_tsv(mf, tmp_path):
    p = tmp_path / "u.data"
    p.write_text("".join(f"{a}\t{b}\t{c:g}\t881250949\n" for a, b, c in ROWS))
    _expect(mf.load_ratings(p))
    _expect(mf.load_ratings(p, "ml-tsv"))


def test_movielens_dat(mf, tmp_path):
    p = tmp_path / "ratings.dat"
    p.write_text("".join(f"{a}::{b}::{c:g}::978300760\n" for a, b, c in ROWS))
    _expect(mf.load_ratings(p))


def test_movielens_csv_with_header_and_crlf(mf, tmp_path):
    p = tmp_path / "ratings.csv"
    p.write_bytes(("userId,movieId,rating,timestamp\r\n" + "".join(f"{a},{b},{c},1112486027\r\n" for a, b, c in ROWS)).encode())
    _expect(mf.load_ratings(p))


def test_netflix_combined(mf, tmp_path):
    p = tmp_path / "combined_data_1.txt"
    by_movie = {}
    for a, b, c in ROWS:
        by_movie.setdefault(b, []).append((a, c))
    txt = ""
    order = []
    for b, lst in by_movie.items():
        txt += f"{b}:\n"
        for a, c in lst:
            txt += f"{a},{int(c) if c == int(c) else c},2005-09-06\n"
            order.append((a, b, c))
    p.write_text(txt)
    d = mf.load_ratings(p)
    uid = sorted({a for a, _, _ in ROWS})
    iid = sorted({b for _, b, _ in ROWS})
    np.testing.assert_array_equal(d["u"], [uid.index(a) for a, _, _ in order])
    np.testing.assert_array_equal(d["i"], [iid.index(b) for _, b, _ in order])
    np.testing.assert_array_equal(d["r"], np.array([c for _, _, c in order], np.float32))


def test_bad_files(mf, tmp_path):
    with pytest.raises(mf.MfsgdError):
        mf.load_ratings(tmp_path / "missing.csv")
    p = tmp_path / "broken.data"
    p.write_text("1\t2\t3.0\t0\n1\tx\t3.0\t0\n")
    with pytest.raises(mf.MfsgdError) as ei:
        mf.load_ratings(p)
    assert "line 2" in str(ei.value)
    p = tmp_path / "empty.csv"
    p.write_text("userId,movieId,rating,timestamp\n")
    d = mf.load_ratings(p, "ml-csv")
    assert d["u"].size == 0 and d["n_users"] == 0


def test_factor_file_roundtrip(mf, tmp_path):
    rng = np.random.default_rng(1)
    P = rng.random((7, 10), dtype=np.float32)
    Q = rng.random((5, 10), dtype=np.float32)
    f = tmp_path / "factors.bin"
    with mf.MatrixFactorizationSGD(7, 5, 10, 0.01, 0.05, 1) as m:
        m.set_factors(P, Q)
        m.save_factors(f)
    assert f.stat().st_size == 8 + 16 + 4 * (70 + 50)
    with mf.MatrixFactorizationSGD(7, 5, 10, 0.01, 0.05, 2) as m:
        m.load_factors(f)
        P2, Q2 = m.get_factors()
    np.testing.assert_array_equal(P, P2)
    np.testing.assert_array_equal(Q, Q2)
    with mf.MatrixFactorizationSGD(7, 5, 8, 0.01, 0.05, 2) as m:
        with pytest.raises(mf.MfsgdError):
            m.load_factors(f)  # k differs
    f.write_bytes(b"not a factor file at all")
    with mf.MatrixFactorizationSGD(7, 5, 10, 0.01, 0.05, 2) as m:
        with pytest.raises(mf.MfsgdError):
            m.load_factors(f)
