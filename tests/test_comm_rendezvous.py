"""CPU: the torch-free one-node rendezvous of comm.py (abstract Unix socket): every rank ends up with rank 0's 128 bytes.
The id itself would come from RCCL; here a stand-in generator is patched in (no GPU, no librccl needed)."""
import importlib
import multiprocessing as mp
import os

import pytest

from helpers import PKG


def _rank(rank, world, tag, q):
    os.environ["R3D_RENDEZVOUS"] = tag
    CM = importlib.import_module(PKG + ".comm")
    CM.Comm.unique_id = staticmethod(lambda: bytes([(rank * 7 + i) % 256 for i in range(CM.ID_BYTES)]))
    try:
        q.put((rank, CM.exchange_unique_id(rank, world, timeout=30.0)))
    except Exception as e:           # pragma: no cover
        q.put((rank, repr(e)))


@pytest.mark.parametrize("world", [1, 2, 5])
def test_every_rank_receives_rank0_id(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    tag = "test_%d_%d" % (os.getpid(), world)
    procs = [ctx.Process(target=_rank, args=(r, world, tag, q)) for r in reversed(range(world))]   # rank 0 starts LAST
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
    want = bytes([i % 256 for i in range(128)])
    assert got == {r: want for r in range(world)}


def test_env_rank_world_and_timeout(monkeypatch):
    CM = importlib.import_module(PKG + ".comm")
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "3")
    assert CM.env_rank_world() == (3, 4)
    monkeypatch.setenv("RANK", "4")
    with pytest.raises(ValueError):
        CM.env_rank_world()
    monkeypatch.setenv("R3D_RENDEZVOUS", "nobody_%d" % os.getpid())
    with pytest.raises(TimeoutError):
        CM.exchange_unique_id(1, 2, timeout=0.3)                    # no rank 0 anywhere


def test_a_rank_whose_first_read_failed_is_served_again(monkeypatch):
    """Rank 0 hands the id to a connection that dies before reading it (no confirmation).  The name must stay up for the
    rank's second attempt instead of vanishing once 'everyone was served' (round-3 advisor finding)."""
    import socket
    import threading
    import time
    CM = importlib.import_module(PKG + ".comm")
    monkeypatch.setenv("R3D_RENDEZVOUS", "retry_%d" % os.getpid())
    monkeypatch.setattr(CM.Comm, "unique_id", staticmethod(lambda: bytes(range(128))))
    out = {}
    t0 = threading.Thread(target=lambda: out.setdefault("r0", CM.exchange_unique_id(0, 2, timeout=20.0)))
    t0.start()
    name = CM._rendezvous_name()
    for _ in range(200):                                   # the failing first attempt of rank 1
        c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        try:
            c.connect(name)
            c.sendall(CM._nonce() + (1).to_bytes(4, "little"))
            c.close()                                      # ... gone before a single byte of the id was read
            break
        except OSError:
            c.close()
            time.sleep(0.02)
    time.sleep(0.3)
    assert CM.exchange_unique_id(1, 2, timeout=10.0) == bytes(range(128))      # the retry finds the name still there
    t0.join(timeout=15)
    assert not t0.is_alive() and out["r0"] == bytes(range(128))
