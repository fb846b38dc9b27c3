"""The int8 MFMA scan of batches is chosen by itself only from 300 k rows on (k <= 32; k <= 128: from 2 M rows
-- css_index.hip: batch_i8_wanted), so the small-index parity suite would hardly reach it: run that whole suite once more in a child process with
CSS_KNN_SCAN=i8 (the switch is read once per process), every batched inner-product search of it on the int8 rows."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
# the passes that keep a REPLACED path under test (switch = "0") leave out the cases whose cost is the CPU oracle and
# whose GPU side the other passes already run: 100 k rows x 1000 queries, 40 k rows x 4200 queries, k beyond one pass
LIGHT = ["-k", "not config2 and not sixteen_query_tiles and not beyond_one_kernel_pass and not two_host_threads"]
# the passes that force a FEW-QUERY path leave out what only batches reach
FEWQ = ["-k", "not config2_100k_768_1000 and not sixteen_query_tiles and not mfma_batch and not no_shadow and not query_chunks "
        "and not two_host_threads and not many_flagged and not searches_of_one_index and not row_widths and not error_band"]


@pytest.mark.gpu
def test_parity_suite_with_the_int8_scan_forced():
    """... once with every later stage on k_scan_qreg_i8 (queries resident in registers: rows of 256 / 512 / 768 padded
    columns) and once with CSS_KNN_QREG=0, which keeps every stage on k_scan_coarse8 (what other row widths use)."""
    # (CSS_KNN_QREG_MIN=0: EVERY later stage on k_scan_qreg_i8, also those of a few tiles; the mix with k_scan_coarse8 that
    # the default minimum gives is what tests/test_fullsize_gpu.py and the bench's self-check run)
    for qreg, qmin in (("1", "0"), ("0", "1024")):
        env = dict(os.environ, CSS_KNN_SCAN="i8", CSS_KNN_QREG=qreg, CSS_KNN_QREG_MIN=qmin)
        # (the third pass only re-runs what reaches the batch scan)
        subset = LIGHT if qreg == "0" else []
        r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_knn_gpu.py"), "-q", "-x", "-m", "gpu",
                            "-p", "no:cacheprovider"] + subset, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
        tail = "\n".join(r.stdout.splitlines()[-15:])
        assert r.returncode == 0, f"test_knn_gpu.py under CSS_KNN_SCAN=i8 CSS_KNN_QREG={qreg} CSS_KNN_QREG_MIN={qmin} failed:\n{tail}\n{r.stderr[-2000:]}"
        assert " passed" in tail


@pytest.mark.gpu
def test_parity_suite_with_the_one_launch_cascade_forced():
    """1..4 queries: the sweep cascade as ONE persistent launch (k_sweep_cascade) is chosen by itself for a single query and, for
    2..4 queries, only from 1 M rows on (css_index.hip: launch_scan_coarse); CSS_KNN_SWEEP_FUSED=2 sends every few-query
    search of the parity suite through it -- masks, L2, ragged sizes, duplicates, k > 128 passes included -- and
    CSS_KNN_SWEEP_FUSED=0 keeps the launch-per-stage cascade it replaces under the same tests."""
    for mode in ("2", "0"):
        env = dict(os.environ, CSS_KNN_SWEEP_FUSED=mode)
        r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_knn_gpu.py"), "-q", "-x", "-m", "gpu",
                            "-p", "no:cacheprovider"] + FEWQ, cwd=str(ROOT), env=env,
                           capture_output=True, text=True, timeout=900)
        tail = "\n".join(r.stdout.splitlines()[-15:])
        assert r.returncode == 0, f"test_knn_gpu.py under CSS_KNN_SWEEP_FUSED={mode} failed:\n{tail}\n{r.stderr[-2000:]}"
        assert " passed" in tail


@pytest.mark.gpu
def test_one_launch_cascade_that_gives_up_still_answers_exactly():
    """k_sweep_cascade bounds every wait: a wave that gives up flags the queries and the device-side exact fix-up answers
    them.  CSS_KNN_FS_SPINS=0 makes every wave give up the first time it would have to wait for a threshold, so the
    few-query searches of the parity suite all end in that fall-back -- slow, and still exact."""
    env = dict(os.environ, CSS_KNN_SWEEP_FUSED="2", CSS_KNN_FS_SPINS="0")
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_knn_gpu.py"), "-q", "-x", "-m", "gpu",
                        "-p", "no:cacheprovider", "-k", "known_answers or k_values or ragged or mask or l2 or duplicate"],
                       cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, f"test_knn_gpu.py with waits that give up at once failed:\n{tail}\n{r.stderr[-2000:]}"
    assert " passed" in tail


@pytest.mark.gpu
def test_parity_suite_with_the_int8_mfma_sweep_forced():
    """3..32 inner-product queries take the int8-MFMA sweep (k_sweep_mfma_i8) by themselves only from 50 k rows on (k <= 32;
    above: from 1 M rows; 17..32 queries: 50 k .. 4 M rows, k > 32 from 2 M rows -- css_index.hip: mfma_sweep_applies); CSS_KNN_SWEEP_MFMA=2 sends every such search of the parity
    suite through it -- tiny and ragged indexes, masks, id bases, duplicates, k = 100 -- and CSS_KNN_SWEEP_MFMA=0 keeps
    what it replaces under the same tests."""
    for mode in ("2", "0"):
        env = dict(os.environ, CSS_KNN_SWEEP_MFMA=mode)
        r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_knn_gpu.py"), "-q", "-x", "-m", "gpu",
                            "-p", "no:cacheprovider"] + FEWQ + ["--deselect",
                            "tests/test_knn_gpu.py::test_three_to_sixteen_queries_sweep_on_the_int8_mfma"],
                           cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
        tail = "\n".join(r.stdout.splitlines()[-15:])
        assert r.returncode == 0, f"test_knn_gpu.py under CSS_KNN_SWEEP_MFMA={mode} failed:\n{tail}\n{r.stderr[-2000:]}"
        assert " passed" in tail
