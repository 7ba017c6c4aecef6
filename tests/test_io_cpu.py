"""Host-side file formats, run bookkeeping and command-line validation (no GPU needed),
checked against vectors produced by the reference (tests/golden/make_golden_io.py)."""
import argparse
import json
import os

import numpy as np
import pytest

from conftest import load_golden

import stereo
import stereo_io
from oracle import oracle as orc


@pytest.fixture(scope="module")
def gold():
    return load_golden("io_metrics")


def test_oracle_to_gray_matches_reference(gold):
    names = [k[len("gray_in_"):] for k in gold if k.startswith("gray_in_")]
    assert len(names) == 9
    for name in names:
        img = gold[f"gray_in_{name}"]
        with np.errstate(invalid="ignore"):
            np.testing.assert_array_equal(orc.to_gray(img), gold[f"gray_raw_{name}"])
            np.testing.assert_array_equal(orc.to_gray(img, normalize=True), gold[f"gray_norm_{name}"])
    np.testing.assert_array_equal(orc.to_gray(gold["gray_in_rgb_u8"], True, 200.0), gold["gray_norm_rgb_u8_max200"])


def test_oracle_metrics_match_reference(gold):
    for thr in (0.5, 1.0, 2.0):
        pbm, rms, n = orc.pbm_rms(gold["metrics_disp"], gold["metrics_gt"], gold["metrics_mask"], thr)
        want = gold[f"metrics_thr{thr}"]
        assert (pbm, rms, float(n)) == (want[0], want[1], want[3])


def test_write_pfm_bytes_match_reference(gold, tmp_path):
    p = str(tmp_path / "g.pfm")
    stereo_io.write_pfm(p, gold["pfm_gray_image"])
    np.testing.assert_array_equal(np.fromfile(p, dtype=np.uint8), gold["pfm_gray_bytes"])
    stereo_io.write_pfm(p, gold["pfm_color_image"], scale=2.5)
    np.testing.assert_array_equal(np.fromfile(p, dtype=np.uint8), gold["pfm_color_bytes"])


def test_read_pfm(gold, tmp_path):
    p = str(tmp_path / "g.pfm")
    gold["pfm_gray_bytes"].tofile(p)
    got = stereo_io.read_pfm(p)
    assert got.dtype == np.float32 and got.flags["C_CONTIGUOUS"]
    np.testing.assert_array_equal(got, gold["pfm_gray_image"])
    gold["pfm_color_bytes"].tofile(p)
    np.testing.assert_array_equal(stereo_io.read_pfm(p), gold["pfm_color_image"])
    gold["pfm_big_endian_bytes"].tofile(p)  # big endian, comment and blank lines in the header
    got = stereo_io.read_pfm(p)
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got, gold["pfm_big_endian_read"])


def test_read_pfm_errors(tmp_path):
    p = str(tmp_path / "bad.pfm")
    with open(p, "wb") as fh:
        fh.write(b"P5\n2 2\n-1.0\n" + bytes(16))
    with pytest.raises(ValueError):
        stereo_io.read_pfm(p)
    with open(p, "wb") as fh:
        fh.write(b"Pf\n2 2\n-1.0\n" + bytes(12))  # one float short
    with pytest.raises(ValueError):
        stereo_io.read_pfm(p)
    with open(p, "wb") as fh:
        fh.write(b"Pf\n2 2\n")
    with pytest.raises(ValueError):
        stereo_io.read_pfm(p)
    with pytest.raises(ValueError):
        stereo_io.write_pfm(p, np.zeros(4, np.float32))
    with pytest.raises(ValueError):
        stereo_io.write_pfm(p, np.zeros((2, 2, 4), np.float32))


def test_pgm_mask_and_read_image(tmp_path):
    from PIL import Image

    m = (np.arange(35, dtype=np.uint8).reshape(5, 7) % 3) * 100
    p = str(tmp_path / "m.pgm")
    Image.fromarray(m).save(p)
    np.testing.assert_array_equal(stereo_io.read_pgm_mask(p), m > 0)
    rgb = np.stack([m, m, m], axis=-1)
    p2 = str(tmp_path / "m.png")
    Image.fromarray(rgb).save(p2)
    np.testing.assert_array_equal(stereo_io.read_image(p2), rgb)
    np.testing.assert_array_equal(stereo_io.read_pgm_mask(p2), m > 0)


def test_npz_helpers(tmp_path):
    d = np.arange(12, dtype=np.int32).reshape(3, 4)
    c = np.linspace(0, 1, 12, dtype=np.float32).reshape(3, 4)
    p = str(tmp_path / "r.npz")
    stereo_io.save_disparity_npz(p, d, c)
    d2, c2 = stereo_io.load_disparity_npz(p)
    np.testing.assert_array_equal(d2, d)
    np.testing.assert_array_equal(c2, c)
    assert d2.dtype == np.int32 and c2.dtype == np.float32
    with pytest.raises(ValueError):
        stereo_io.save_disparity_npz(p, d, c[:2])
    with pytest.raises(ValueError):
        stereo_io.save_npz(p, {})
    stereo_io.save_npz(p, {"x": d})
    with pytest.raises(ValueError):
        stereo_io.load_disparity_npz(p)
    # converters
    stereo_io.save_disparity_npz(p, d, c)
    pfm = str(tmp_path / "r.pfm")
    stereo_io.convert_npz_to_pfm(p, pfm)
    np.testing.assert_array_equal(stereo_io.read_pfm(pfm), d.astype(np.float32))
    with pytest.raises(ValueError):
        stereo_io.convert_npz_to_pfm(p, pfm, key="nope")
    back = str(tmp_path / "b.npz")
    stereo_io.convert_pfm_to_npz(pfm, back, key="gt")
    np.testing.assert_array_equal(stereo_io.load_npz(back)["gt"], d.astype(np.float32))


def test_to_gray_shape_and_scale_validation():
    with pytest.raises(ValueError):
        stereo_io.to_gray(np.zeros((4, 4, 2), np.uint8))
    with pytest.raises(ValueError):
        stereo_io.to_gray(np.zeros(4, np.uint8))
    with pytest.raises(ValueError):
        stereo_io.to_gray(np.zeros((4, 4), np.uint8), normalize=True, max_value=0.0)
    assert stereo_io._resolve_normalization_scale(np.zeros((2, 2), np.uint16), None) == 65535.0
    assert stereo_io._resolve_normalization_scale(np.zeros((2, 2), np.float32), 7) == 7.0
    out = stereo_io.to_gray(np.zeros((0, 5, 3), np.uint8))
    assert out.shape == (0, 5) and out.dtype == np.float32


def test_jet_colormap_matches_reference(gold):
    got = stereo._jet_colormap(gold["jet_in"])
    assert got.dtype == gold["jet_out"].dtype
    np.testing.assert_array_equal(got, gold["jet_out"])
    with pytest.raises(ValueError):
        stereo._jet_colormap(np.zeros(3, np.float32))


def test_disparity_images(tmp_path):
    from PIL import Image

    d = np.arange(64, dtype=np.int32).reshape(8, 8)
    stereo._save_disparity_image(d, 64, str(tmp_path / "g.png"))
    g = np.array(Image.open(tmp_path / "g.png"))
    np.testing.assert_array_equal(g, ((d.astype(np.float32) / 63.0) * 255.0).astype(np.uint8))
    stereo._save_disparity_color_image(d, 64, str(tmp_path / "c.png"))
    c = np.array(Image.open(tmp_path / "c.png"))
    want = (stereo._jet_colormap(d.astype(np.float32) / 63.0) * 255.0).astype(np.uint8)
    np.testing.assert_array_equal(c, want)
    with pytest.raises(ValueError):
        stereo._save_disparity_image(d, 0, str(tmp_path / "x.png"))
    with pytest.raises(ValueError):
        stereo._save_disparity_color_image(d, 0, str(tmp_path / "x.png"))


def test_run_directory_naming(tmp_path):
    base = str(tmp_path / "result")
    a = stereo._create_run_directory(base, "20240102030405")
    b = stereo._create_run_directory(base, "20240102030405")
    c = stereo._create_run_directory(base, "20240102030405")
    assert a.name == "20240102030405" and b.name == "20240102_01_030405" and c.name == "20240102_02_030405"
    assert all(p.is_dir() for p in (a, b, c))
    e = stereo._create_run_directory(base, "030405")
    f = stereo._create_run_directory(base, "030405")
    assert e.name == "030405" and f.name == "01_030405"


def test_ndisp_summary(tmp_path):
    p = tmp_path / "ndisp_summary.csv"
    with pytest.raises(FileNotFoundError):
        stereo._read_ndisp_summary(p)
    p.write_text("scene,ndisp,width\nAdirondack-perfect,290,2880\nAdirondack-imperfect,290,2880\n"
                 "Broken-perfect,abc,1\n,100,1\nPiano-perfect, 260 ,2820\nEmpty-perfect,,1\n", encoding="utf-8")
    assert stereo._read_ndisp_summary(p) == {"Adirondack-perfect": 290, "Piano-perfect": 260}
    p.write_text("scene,ndisp\nX-imperfect,10\n", encoding="utf-8")
    with pytest.raises(ValueError):
        stereo._read_ndisp_summary(p)


def _args(**kw):
    base = dict(left="", right="", dmax=0, dataset="", all_datasets=False, gt="", gt_mask="", wct_radius=4,
                base_weight=8.0, guided_radius=3, guided_eps=0.01, filter="guided", median_radius=3,
                median_method="opencv", median_block_rows=128, gaussian_sigma=1.0, bilateral_sigma=1.0,
                bad_threshold=2.0)
    base.update(kw)
    return argparse.Namespace(**base)


def test_parse_args_defaults():
    a = stereo._parse_args([])
    assert vars(a) == vars(_args())
    a = stereo._parse_args(["--dataset", "Piano-perfect", "--filter", "median", "--gt-mask", "m.pgm", "--all-datasets"])
    assert a.dataset == "Piano-perfect" and a.filter == "median" and a.gt_mask == "m.pgm" and a.all_datasets
    with pytest.raises(SystemExit):
        stereo._parse_args(["--filter", "box"])


def test_validate_args(tmp_path):
    root = tmp_path / "dataset"
    nd = {"A-perfect": 32}
    bad = [
        _args(all_datasets=True, dataset="A-perfect"),
        _args(all_datasets=True, left="l.png"),
        _args(all_datasets=True, dmax=5),
        _args(dataset="A-perfect", gt="x.pfm"),
        _args(dataset="A-perfect", dmax=-1),
        _args(),
        _args(left="l", right="r"),
        _args(left="l", right="r", dmax=8),
        _args(left="l", right="r", dmax=8, gt="g"),
    ]
    root.mkdir()
    for a in bad:
        with pytest.raises(ValueError):
            stereo._validate_args(a, root, nd)
    with pytest.raises(ValueError):
        stereo._validate_args(_args(dataset="B-perfect"), root, nd)
    with pytest.raises(ValueError):
        stereo._validate_args(_args(all_datasets=True), root, {})
    with pytest.raises(FileNotFoundError):
        stereo._validate_args(_args(all_datasets=True), tmp_path / "missing", nd)
    with pytest.raises(FileNotFoundError):
        stereo._validate_args(_args(dataset="A-perfect"), tmp_path / "missing", nd)
    stereo._validate_args(_args(all_datasets=True), root, nd)
    stereo._validate_args(_args(dataset="A-perfect"), root, nd)
    stereo._validate_args(_args(dataset="B-perfect", dmax=16), root, nd)
    stereo._validate_args(_args(left="l", right="r", dmax=8, gt="g", gt_mask="m"), root, nd)


def test_scene_inputs(tmp_path):
    root = tmp_path / "dataset"
    scene = root / "A-perfect"
    scene.mkdir(parents=True)
    names = ("im0.png", "im1.png", "disp0.pfm", "disp0-n.pgm")
    for i, n in enumerate(names):
        with pytest.raises(FileNotFoundError):
            stereo._resolve_scene_inputs(root, "A-perfect", {"A-perfect": 32}, 0)
        (scene / n).write_bytes(b"x")
    got = stereo._resolve_scene_inputs(root, "A-perfect", {"A-perfect": 32}, 0)
    assert got == (*(scene / n for n in names), 32)
    assert stereo._resolve_scene_inputs(root, "A-perfect", {}, 48)[4] == 48
    with pytest.raises(ValueError):
        stereo._resolve_scene_inputs(root, "A-perfect", {}, 0)


def test_run_metadata(tmp_path):
    from pathlib import Path

    run = tmp_path / "20240102030405"
    common = {"timestamp", "run_dir", "left", "right", "dmax", "wct_radius", "base_weight", "filter", "bad_threshold",
              "eval", "gt", "gt_mask", "dataset", "all_datasets", "progress", "output_disparity_png",
              "output_disparity_color_png", "output_disparity_npz", "output_metrics_json"}
    extra = {"guided": {"guided_radius", "guided_eps"}, "median": {"median_radius", "median_method", "median_block_rows"},
             "gaussian": {"gaussian_sigma"}, "bilateral": {"bilateral_sigma"}}
    for filt, keys in extra.items():
        meta = stereo._build_run_metadata(_args(filter=filt), run, run / "g.png", run / "c.png", run / "d.npz", None,
                                          Path("l.png"), Path("r.png"), Path("gt.pfm"), Path("m.pgm"), 64, "S", True)
        assert set(meta) == common | keys
        assert all(isinstance(v, str) for v in meta.values())
        assert meta["output_metrics_json"] == "" and meta["dmax"] == "64" and meta["all_datasets"] == "True"
        assert meta["eval"] == "True" and meta["timestamp"] == run.name
    p = tmp_path / "params.json"
    stereo._write_run_metadata(p, meta)
    text = p.read_text()
    assert json.loads(text) == meta and text.startswith("{\n  \"all_datasets\"")


def test_metrics_argument_validation():
    d = np.zeros((4, 5), np.int32)
    g = np.zeros((4, 5), np.float32)
    m = np.ones((4, 5), bool)
    for args in ((d, g[:3], m, 1.0), (d[0], g[0], m[0], 1.0), (d, g, m[:3], 1.0), (d, g, m, 0.0)):
        with pytest.raises(ValueError):
            stereo._compute_pbm_rms(*args)


def test_progress_printers(capsys):
    stereo._print_progress(3, 4, "X")
    stereo._print_progress(9, 4, "X")
    state = stereo.DatasetProgressState()
    stereo._print_dataset_stage_progress(1, 2, 5, 10, "Stage", state)
    assert state.rendered
    stereo._print_dataset_stage_progress(2, 2, 10, 10, "Stage", state)
    out = capsys.readouterr().out
    assert "X: 3/4 ( 75.0%)" in out and "X: 4/4 (100.0%)\n" in out
    assert "Dataset: 1/2 ( 50.0%)\nStage: 5/10 ( 50.0%)" in out and "\033[1A\r\033[2KDataset: 2/2 (100.0%)" in out
    with pytest.raises(ValueError):
        stereo._print_progress(1, 0, "X")
    with pytest.raises(ValueError):
        stereo._print_dataset_stage_progress(1, 0, 1, 1, "X", state)
    with pytest.raises(ValueError):
        stereo._print_dataset_stage_progress(1, 1, 1, 0, "X", state)
    stereo._print_metrics({"pbm": 12.34567, "rms": 1.5, "bad_threshold": 2.0, "valid_pixel_count": 100.0})
    out = capsys.readouterr().out
    assert "PBM: 12.3457%" in out and "RMS: 1.5000" in out and "Valid Pixel Count: 100" in out


def test_convert_tool(tmp_path):
    import convert

    d = np.arange(20, dtype=np.float32).reshape(4, 5)
    npz, pfm, back = str(tmp_path / "a.npz"), str(tmp_path / "a.pfm"), str(tmp_path / "b.npz")
    stereo_io.save_npz(npz, {"disparity": d})
    convert.main(["--input", npz, "--output", pfm, "--mode", "npz2pfm"])
    np.testing.assert_array_equal(stereo_io.read_pfm(pfm), d)
    convert.main(["--input", pfm, "--output", back, "--mode", "pfm2npz", "--key", "gt"])
    np.testing.assert_array_equal(stereo_io.load_npz(back)["gt"], d)
    with pytest.raises(SystemExit):
        convert.main(["--input", npz, "--output", pfm, "--mode", "other"])
