"""The host-buffer pipeline of the C ABI (vs_synth, vs_synth_rows): chunked compute, pinned
staging, four delivery workers.  Whatever the route, the bytes must equal the one-launch result
of the device-pointer path (vs_plan_launch + one download), which the parity tests pin."""
import threading

import numpy as np
import pytest

import voice_synth_amd as vs
from voice_synth_amd import configs
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _one_shot(engine, lanes, ns):
    """device-pointer path: one plan, one launch, one download"""
    n = len(lanes)
    pitch = (ns + 7) & ~7
    plan = engine.plan(lanes, ns)
    out = engine.dev_alloc(n * pitch * 2)
    plan.launch(vs.VS_KIND_SYNTH, out, out_pitch=pitch)
    plan.status()
    pcm = engine.dev_download(out, (n, pitch))[:, :ns].copy()
    engine.dev_free(out)
    plan.close()
    return pcm


@pytest.fixture(scope="module")
def batch(engine):
    # 40000 utterances x 3000 samples: three compute chunks (16384 + 16384 + 7232), 15 staging blocks
    specs, fs, dur, _ = configs.config_specs(3, 40000)
    lanes, d = vs.lanes_from_specs(specs)
    ns = 3000
    return lanes, ns, _one_shot(engine, lanes, ns)


def test_pageable_destination_equals_one_shot(engine, batch):
    lanes, ns, want = batch
    got = engine.synth(lanes, ns)
    assert np.array_equal(got, want)
    pick = [0, 16383, 16384, 32767, 32768, 39999]
    assert np.array_equal(got[pick], po.synth([lanes[i] for i in pick], ns))


def test_pinned_destination_equals_one_shot(engine, batch):
    lanes, ns, want = batch
    got = engine.synth_pinned(lanes, ns)
    try:
        assert np.array_equal(got, want)
    finally:
        engine.host_free(got)


def test_rows_callback_delivers_every_row_exactly_once(engine, batch):
    lanes, ns, want = batch
    seen = np.zeros(len(lanes), dtype=np.int32)
    bad = []
    lock = threading.Lock()

    def fn(row0, rows):
        ok = np.array_equal(rows, want[row0:row0 + rows.shape[0]])
        with lock:
            seen[row0:row0 + rows.shape[0]] += 1
            if not ok:
                bad.append(row0)
        return 0

    engine.synth_rows(lanes, ns, fn)
    assert not bad, bad[:5]
    assert (seen == 1).all()


def test_callback_failure_stops_the_pipeline(engine, batch):
    lanes, ns, _ = batch
    with pytest.raises(vs.VsError) as e:
        engine.synth_rows(lanes, ns, lambda row0, rows: 1)
    assert e.value.code == vs._ffi.VS_ERR_IO
    # the context is usable afterwards
    small = engine.synth(lanes[:70], ns)
    assert np.array_equal(small, po.synth(lanes[:70], ns))


def test_ragged_last_chunk_and_odd_sample_count(engine):
    specs, fs, dur, _ = configs.config_specs(2, 16384 + 17)
    lanes, d = vs.lanes_from_specs(specs)
    ns = 1001
    got = engine.synth(lanes, ns)
    pick = [0, 5, 16383, 16384, 16400]
    assert np.array_equal(got[pick], po.synth([lanes[i] for i in pick], ns))
    assert np.array_equal(got, _one_shot(engine, lanes, ns))


def test_trim_releases_and_the_context_keeps_working(engine):
    engine.trim()
    specs, fs, dur, _ = configs.config_specs(3, 100)
    lanes, d = vs.lanes_from_specs(specs)
    assert np.array_equal(engine.synth(lanes, 2000), po.synth(lanes, 2000))


def test_a_row_longer_than_a_staging_block(engine):
    """utterances of more than 8.4 M samples do not fit a 16 MiB staging block: the context then
    allocates staging buffers that hold a whole row (the callback contract is whole rows)"""
    specs, fs, dur, _ = configs.config_specs(3, 3)
    lanes, d = vs.lanes_from_specs(specs)
    n = 8_500_001
    want = po.synth(lanes, n)
    assert np.array_equal(engine.synth(lanes, n), want)
    got = np.zeros_like(want)

    def sink(row0, block):
        got[row0:row0 + len(block)] = block
        return 0
    engine.synth_rows(lanes, n, sink)
    assert np.array_equal(got, want)
    engine.trim()                     # give the 4 x 17 MiB of pinned memory back
    assert np.array_equal(engine.synth(lanes[:2], 1000), want[:2, :1000])
