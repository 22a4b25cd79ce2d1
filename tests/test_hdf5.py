"""CPU: the slide-grid HDF5 format.  The writer is checked against libhdf5 itself
(h5py under /opt/conda's interpreter, `h5dump`) when the image has it, and against this package's own
reader otherwise; the reader is checked on a file h5py wrote (the reference's converter, convert.py:27-32,
is h5py's `create_group` + `create_dataset(stem, data=array)`)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import load_golden

CONDA_PY = "/opt/conda/bin/python3.9"
H5DUMP = "/opt/conda/bin/h5dump"


def _have_h5py():
    if not os.path.exists(CONDA_PY):
        return False
    r = subprocess.run([CONDA_PY, "-c", "import h5py"], capture_output=True, cwd="/tmp", env={"PATH": "/usr/bin:/bin"})
    return r.returncode == 0


def _groups(n_extra=0):
    rng = np.random.default_rng(0)
    g = load_golden("driver")
    groups = {
        "images": {"slide_a": g["grid:images/slide_a"],                                   # uint8 code grid
                   "slide_b": g["grid:images/slide_b"],                                   # bool (codes {0, 1})
                   "tumor_003": rng.integers(0, 1024, (33, 47)).astype(np.uint16),        # K = 1024 codebook
                   "wide": rng.integers(-3, 70000, (5, 6)).astype(np.int32),
                   "ref_dtype": rng.integers(0, 256, (4, 4)).astype(np.int64),            # uncast reference dtype
                   "empty": np.zeros((0, 16), np.uint8)},
        "masks": {"slide_a_mask": rng.integers(0, 2, (24, 20)).astype(bool),
                  "slide_b_mask": rng.integers(0, 3, (12, 16)).astype(np.uint8),
                  "tumor_003_mask": np.zeros((33, 47), bool)},
    }
    for i in range(n_extra):
        groups["images"][f"normal_{i:05d}"] = rng.integers(0, 256, (3, 1 + i % 7)).astype(np.uint8)
    return groups


def _same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)


def test_writer_reader_roundtrip(amd, tmp_path):
    from vqae_amd import hdf5
    groups = _groups(n_extra=2100)                      # > 1024 entries: several symbol-table nodes
    path = hdf5.write_hdf5(tmp_path / "enc.hdf5", groups)
    back = hdf5.read_hdf5(path)
    assert set(back) == {"images", "masks"}
    for g, dsets in groups.items():
        assert sorted(back[g]) == sorted(dsets)
        for n, a in dsets.items():
            assert _same(back[g][n], a), (g, n)
    r = hdf5.H5Reader(path)                             # the downstream dataset's access pattern
    assert _same(r["images"]["slide_a"], groups["images"]["slide_a"])
    assert _same(r["masks"]["slide_a" + "_mask"], groups["masks"]["slide_a_mask"])
    assert "slide_b" in r["images"] and len(r["masks"]) == 3


def test_writer_rejects_bad_input(amd, tmp_path):
    from vqae_amd import hdf5
    w = hdf5.H5Writer(tmp_path / "x.hdf5")
    w.create_dataset("images", "a", np.zeros((2, 2), np.uint8))
    with pytest.raises(ValueError):
        w.create_dataset("images", "a", np.zeros((2, 2), np.uint8))
    with pytest.raises(TypeError):
        w.create_dataset("images", "c", np.zeros((2,), np.complex64))
    with pytest.raises(ValueError):
        w.create_group("a/b")
    w.close()
    with pytest.raises(ValueError):
        w.create_dataset("images", "d", np.zeros((1,), np.uint8))
    (tmp_path / "bad.hdf5").write_bytes(b"not hdf5" * 20)
    with pytest.raises(ValueError):
        hdf5.H5Reader(tmp_path / "bad.hdf5")


def test_convert_npy_to_hdf5_layout(amd, tmp_path):
    """encodings/{images,masks}/<stem>.npy -> groups/datasets (convert.py:27-32)."""
    from vqae_amd.extract_embeddings import convert_npy_to_hdf5
    from vqae_amd import hdf5
    groups = _groups()
    for g, dsets in groups.items():
        (tmp_path / "encodings" / g).mkdir(parents=True)
        for n, a in dsets.items():
            np.save(tmp_path / "encodings" / g / (n + ".npy"), a)
    out = convert_npy_to_hdf5(tmp_path / "encodings")
    assert out == str(tmp_path / "encodings.hdf5")
    back = hdf5.read_hdf5(out)
    assert all(_same(back[g][n], a) for g, d in groups.items() for n, a in d.items())


@pytest.mark.skipif(not _have_h5py(), reason="no libhdf5/h5py interpreter in this image")
def test_libhdf5_reads_our_files_and_we_read_h5py_files(amd, tmp_path):
    from vqae_amd import hdf5
    groups = _groups(n_extra=1500)
    ours = hdf5.write_hdf5(tmp_path / "ours.hdf5", groups)
    flat = {f"{g}/{n}": a for g, d in groups.items() for n, a in d.items()}
    np.savez(tmp_path / "want.npz", **flat)
    script = f"""
import h5py, numpy as np, warnings
warnings.simplefilter('ignore')
z = np.load(r'{tmp_path}/want.npz')
with h5py.File(r'{ours}', 'r') as f:                     # libhdf5 parses our file
    assert sorted(f.keys()) == ['images', 'masks']
    for k in z.files:
        a = np.asarray(f[k.split('/')[0]][k.split('/')[1]])
        assert a.dtype == z[k].dtype and a.shape == z[k].shape and np.array_equal(a, z[k]), k
    assert len(f['images']) == {len(groups['images'])}
with h5py.File(r'{tmp_path}/theirs.hdf5', 'w') as f:     # what the reference's converter does
    for k in z.files:
        g, n = k.split('/')
        f.require_group(g).create_dataset(n, data=z[k])
print('OK')
"""
    r = subprocess.run([CONDA_PY, "-c", script], capture_output=True, text=True, cwd="/tmp", env={"PATH": "/usr/bin:/bin"})
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]
    theirs = hdf5.read_hdf5(tmp_path / "theirs.hdf5")
    for g, d in groups.items():
        for n, a in d.items():
            assert _same(theirs[g][n], a), (g, n)
    if os.path.exists(H5DUMP):
        d = subprocess.run([H5DUMP, "-H", "-d", "/masks/slide_a_mask", ours], capture_output=True, text=True)
        assert d.returncode == 0 and '"TRUE"' in d.stdout and "H5T_STD_I8LE" in d.stdout
