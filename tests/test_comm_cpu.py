"""CPU: the rendezvous of dpe_comm (csrc/dpe_comm.hip) through a directory that earlier runs have used.  Only the handshake
runs here (host-file backend: nothing touches a GPU until the first exchange); the exchanges themselves are GPU tests
(tests/test_gpu_comm.py)."""
import os
import struct
import threading
import time

import pytest

import navlab_dpe_sdr_amd as dpe


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as ge
    ge.build()


def _create_all(path, n, timeout=30.0):
    out, err = [None] * n, [None] * n

    def run(r):
        try:
            out[r] = dpe.engine.Comm(r, n, str(path), dpe.engine.Comm.HOSTFILES)
        except Exception as e:      # noqa: BLE001 -- reported to the asserting thread
            err[r] = e

    th = [threading.Thread(target=run, args=(r,)) for r in range(n)]
    for t in reversed(th):      # ranks > 0 first: rank 0 must cope with peers that are already waiting
        t.start()
        time.sleep(0.01)
    for t in th:
        t.join(timeout)
    assert not any(t.is_alive() for t in th), "rendezvous hung"
    assert err == [None] * n, err
    return out


def test_rendezvous_in_a_fresh_and_in_a_reused_directory(tmp_path):
    for _ in range(3):      # same directory three times: nothing an earlier run left behind may be believed
        comms = _create_all(tmp_path, 3)
        for c in comms:
            c.close()
    assert not [f for f in os.listdir(tmp_path) if f.startswith(("rendezvous", "join.", "ack."))]   # rank 0 cleaned up


def test_stale_files_of_a_crashed_run_are_ignored(tmp_path):
    # what a run that died mid-handshake leaves: a rendezvous file with other tokens (and another unique id), an
    # acknowledgement, a join file, and the old-format nccl_id
    stale = struct.pack("<QQq", 0x4450455f52445a31, 0x1111, 2) + struct.pack("<64Q", *([0x2222] * 64)) + bytes(128)
    (tmp_path / "rendezvous").write_bytes(stale)
    (tmp_path / "ack.1").write_bytes(struct.pack("<QQ", 0x1111, 0x2222))
    (tmp_path / "join.1").write_bytes(struct.pack("<Q", 0x2222))
    (tmp_path / "nccl_id").write_bytes(bytes(128))
    comms = _create_all(tmp_path, 2)
    for c in comms:
        c.close()


def test_a_lone_rank_times_out_with_a_message(tmp_path, monkeypatch):
    monkeypatch.setenv("DPE_COMM_TIMEOUT_S", "0.5")
    with pytest.raises(dpe.DpeError, match="no rendezvous for this run"):
        dpe.engine.Comm(1, 2, str(tmp_path), dpe.engine.Comm.HOSTFILES)
    with pytest.raises(dpe.DpeError, match="not every rank joined"):
        dpe.engine.Comm(0, 2, str(tmp_path / "other"), dpe.engine.Comm.HOSTFILES)
