"""The node-level entry of the C ABI (vs_node_*): contiguous blocks of lanes per shard, finished
chunks gathered into the root device by peer copies overlapped with the next chunk's kernel.
With one GPU the shards are LOGICAL (the same device listed N times): the placement logic, the
chunk pipeline and -- with VS_NODE_STAGE_ALL -- the transfer path all run, and the result must be
byte for byte what one shard gives ("multi-GPU without 8 GPUs")."""
import threading

import numpy as np
import pytest

import voice_synth_amd as vs
from voice_synth_amd import _ffi, configs
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def batch(engine):
    specs, fs, dur, _ = configs.config_specs(3, 50000)   # 8 shards of 6250: ragged groups of 64, chunks of one
    lanes, d = vs.lanes_from_specs(specs)
    ns = 2500
    return lanes, ns, engine.synth(lanes, ns)


@pytest.mark.parametrize("shards,flags", [(1, vs.Node.OVERLAP), (8, vs.Node.OVERLAP),
                                          (8, vs.Node.OVERLAP | vs.Node.STAGE_ALL), (3, vs.Node.STAGE_ALL),
                                          (2, vs.Node.OVERLAP | vs.Node.STAGE_ALL)])
def test_gather_into_root_equals_one_device(engine, batch, shards, flags):
    lanes, ns, want = batch
    n = len(lanes)
    pitch = ns + 12                      # a destination pitch of the caller's choosing
    root = engine.dev_alloc(n * pitch * 2)
    node = vs.Node([0] * shards)
    try:
        tot, comp = node.synth_gather(lanes, ns, root, pitch, flags)
        got = engine.dev_download(root, (n, pitch))[:, :ns]
        assert np.array_equal(got, want), (shards, flags)
        assert tot >= comp > 0
    finally:
        node.close()
        engine.dev_free(root)


def test_shard_ranges_are_contiguous_blocks():
    node = vs.Node([0] * 8)
    try:
        edges = [node.shard_range(262144, s) for s in range(8)]
        assert edges == [(s * 32768, (s + 1) * 32768) for s in range(8)]     # BASELINE config 4
        # ragged: blocks differ by at most one lane, exactly as the one-process-per-GPU path cuts them
        from voice_synth_amd.configs import shard_range
        for n in (10, 3, 65537, 262143):
            assert [node.shard_range(n, s) for s in range(8)] == [shard_range(n, s, 8) for s in range(8)]
        assert node.shard_range(10, 0) == (0, 2) and node.shard_range(10, 2) == (4, 5) and node.shard_range(10, 7) == (9, 10)
    finally:
        node.close()


def test_node_rows_to_host_equals_one_device(engine, batch):
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

    node = vs.Node([0] * 4)
    try:
        node.synth_rows(lanes, ns, fn)
    finally:
        node.close()
    assert not bad and (seen == 1).all()


def test_more_shards_than_lanes(engine):
    specs, fs, dur, _ = configs.config_specs(3, 5)
    lanes, d = vs.lanes_from_specs(specs)
    root = engine.dev_alloc(5 * 2000 * 2)
    node = vs.Node([0] * 8)
    try:
        node.synth_gather(lanes, 2000, root, 2000, vs.Node.OVERLAP | vs.Node.STAGE_ALL)
        assert np.array_equal(engine.dev_download(root, (5, 2000)), po.synth(lanes, 2000))
    finally:
        node.close()
        engine.dev_free(root)


def test_links_and_the_rccl_transport_on_one_device(engine, batch):
    """What one GPU can show of the RCCL transport: the node opens librccl, creates a communicator of its
    own (one rank), gathers through that code path, and gives the communicator back; logical shards of one
    device are refused (RCCL puts one rank on one device), and a pitched root buffer too (RCCL messages
    are contiguous).  set_transport() itself ends with the node's link check: 64 KiB from the root's communicator to
    itself through ncclGroupStart / ncclRecv / ncclSend / ncclGroupEnd, compared byte for byte -- so the entry points the
    gather's exchange is made of DO run against the real librccl here (a check that fails raises, and rccl_ranks says the
    communicator stands).  Sends and receives between DIFFERENT devices stay unexercised until a multi-GPU node runs them
    (DESIGN.md section 7)."""
    lanes, ns, want = batch
    node = vs.Node([0])
    try:
        assert node.link(0) == "self"
        node.set_transport(vs.Node.TRANSPORT_RCCL)     # raises if the link check fails
        assert node.rccl_ranks(0) == 1 and node.last_rccl_error() == 0
        root = engine.dev_alloc(len(lanes) * ns * 2)
        try:
            node.synth_gather(lanes, ns, root, ns)
            assert np.array_equal(engine.dev_download(root, (len(lanes), ns), np.int16), want)
            with pytest.raises(vs.VsError) as e:
                node.synth_gather(lanes, ns, root, ns + 8)
            assert e.value.code == _ffi.VS_ERR_UNSUPPORTED
        finally:
            engine.dev_free(root)
        node.set_transport(vs.Node.TRANSPORT_PEER)
    finally:
        node.close()
    node = vs.Node([0, 0, 0])
    try:
        assert [node.link(s) for s in range(3)] == ["self", "self", "self"]
        with pytest.raises(vs.VsError) as e:
            node.set_transport(vs.Node.TRANSPORT_RCCL)
        assert e.value.code == _ffi.VS_ERR_UNSUPPORTED
    finally:
        node.close()


SENTINEL = 0x5A5A


def _fill(engine, ptr, n_words):
    engine.dev_upload(ptr, np.full(n_words, SENTINEL, dtype=np.uint16).view(np.int16))


def test_a_shard_that_cannot_prepare_stops_every_shard(engine, batch):
    """The exchange is all or nothing: the shard threads meet after ALL chunk plans of ALL shards exist; one shard that
    failed keeps every shard from enqueueing anything (no kernel, no copy, no send, no receive) -- an unmatched
    ncclSend / ncclRecv would never complete.  vs_tuning.fault on one shard's context provokes it: the call returns
    that shard's error and not one word of the root buffer has been written."""
    lanes, ns, want = batch
    n = len(lanes)
    root = engine.dev_alloc(n * ns * 2)
    node = vs.Node([0] * 4)
    try:
        _fill(engine, root, n * ns)
        node.set_shard_tuning(2, fault=vs.VS_FAULT_SHARD_PREPARE)
        with pytest.raises(vs.VsError) as e:
            node.synth_gather(lanes, ns, root, ns, vs.Node.OVERLAP | vs.Node.STAGE_ALL)
        assert e.value.code == _ffi.VS_ERR_INTERNAL
        got = engine.dev_download(root, (n, ns)).view(np.uint16)
        assert (got == SENTINEL).all()
        node.set_shard_tuning(2)
        node.synth_gather(lanes, ns, root, ns, vs.Node.OVERLAP | vs.Node.STAGE_ALL)
        assert np.array_equal(engine.dev_download(root, (n, ns)), want)
    finally:
        node.close()
        engine.dev_free(root)


@pytest.mark.parametrize("shard", [0, 3])
def test_a_shard_that_fails_at_its_first_handover_ends_the_call(engine, batch, shard):
    """A failure BEHIND the meeting point (a launch or a send refused): the failing thread ends the exchange, every
    other shard stops at its next chunk, the call returns the error instead of waiting for transfers that will never
    be matched, and the node works again afterwards."""
    lanes, ns, want = batch
    n = len(lanes)
    root = engine.dev_alloc(n * ns * 2)
    node = vs.Node([0] * 4)
    try:
        node.set_shard_tuning(shard, fault=vs.VS_FAULT_SHARD_HANDOVER)
        with pytest.raises(vs.VsError) as e:
            node.synth_gather(lanes, ns, root, ns, vs.Node.OVERLAP | vs.Node.STAGE_ALL)
        assert e.value.code == _ffi.VS_ERR_INTERNAL
        node.set_shard_tuning(shard)
        node.synth_gather(lanes, ns, root, ns, vs.Node.OVERLAP | vs.Node.STAGE_ALL)
        assert np.array_equal(engine.dev_download(root, (n, ns)), want)
    finally:
        node.close()
        engine.dev_free(root)


def test_rccl_exchange_aborted_on_failure_falls_back_to_the_peer_transport(engine, batch):
    """What one GPU can show of the abort path: with the RCCL transport chosen, a failure behind the meeting point calls
    ncclCommAbort on the node's communicator(s); the node is back on the peer transport afterwards, a new communicator
    can be made, and the gather through it is right again."""
    lanes, ns, want = batch
    n = len(lanes)
    root = engine.dev_alloc(n * ns * 2)
    node = vs.Node([0])
    try:
        node.set_transport(vs.Node.TRANSPORT_RCCL)
        node.set_shard_tuning(0, fault=vs.VS_FAULT_SHARD_HANDOVER)
        with pytest.raises(vs.VsError) as e:
            node.synth_gather(lanes, ns, root, ns)
        assert e.value.code == _ffi.VS_ERR_INTERNAL
        node.set_shard_tuning(0)
        # back on the peer transport: a pitched root buffer (which RCCL refuses) is accepted again
        pitched = engine.dev_alloc(n * (ns + 8) * 2)
        try:
            node.synth_gather(lanes, ns, pitched, ns + 8)
            assert np.array_equal(engine.dev_download(pitched, (n, ns + 8))[:, :ns], want)
        finally:
            engine.dev_free(pitched)
        node.set_transport(vs.Node.TRANSPORT_RCCL)
        node.synth_gather(lanes, ns, root, ns)
        assert np.array_equal(engine.dev_download(root, (n, ns)), want)
    finally:
        node.close()
        engine.dev_free(root)
