"""The id-file rendezvous of cpecan_em_run / cpecan_em_comm_create at world > 1 (csrc/cpecan_em.hip), driven with two
processes and no GPU up to -- not including -- ncclCommInitRank: rank 0 publishes bytes, the other rank receives exactly
them; a file an earlier run left at the path (another nonce, or no header at all) is not taken for this rendezvous; a
second rendezvous at the same path in the same job gets its own nonce; rank 0 removes the file afterwards."""
import ctypes as C
import multiprocessing as mp
import os
import struct
import time

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "cpecan-signal_amd", "libcpecan_em.so")


def _lib():
    L = C.CDLL(LIB)
    L.cpecan_em_rendezvous_exchange.restype = C.c_int
    L.cpecan_em_rendezvous_exchange.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int32]
    L.cpecan_em_rendezvous_done.argtypes = [C.c_char_p, C.c_int]
    L.cpecan_em_set_rendezvous_nonce.argtypes = [C.c_uint64]
    L.cpecan_em_last_error.restype = C.c_char_p
    return L


def _rank(rank, path, nonce, delay, rounds, q):
    time.sleep(delay)
    L = _lib()
    L.cpecan_em_set_rendezvous_nonce(nonce)
    got = []
    for r in range(rounds):
        buf = (C.c_ubyte * 128)(*([(17 * r + i) & 255 for i in range(128)] if rank == 0 else [0] * 128))
        rc = L.cpecan_em_rendezvous_exchange(path.encode(), rank, 2, buf, 128, 5000)
        got.append((rc, bytes(buf)))
        if rank == 0:
            # (ncclCommInitRank would return here, once the other rank has joined: stand in for it)
            while os.path.exists(path + ".got%d" % r) is False and time.time() < q[1]:
                time.sleep(0.01)
            L.cpecan_em_rendezvous_done(path.encode(), 0)
        else:
            open(path + ".got%d" % r, "w").close()
    q[0].put((rank, got))


def _run(tmp_path, stale, nonce=12345, rounds=1, delays=(0.3, 0.0)):
    path = str(tmp_path / "rccl.id")
    if stale is not None:
        open(path, "wb").write(stale)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    deadline = time.time() + 30
    ps = [ctx.Process(target=_rank, args=(r, path, nonce, delays[r], rounds, (q, deadline))) for r in range(2)]
    for p in ps:
        p.start()
    out = dict(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join(30)
    return path, out


def _stale(nonce, fill):
    return b"CPECANID" + struct.pack("<Qii", nonce, 2, 128) + bytes([fill] * 128)


@pytest.mark.parametrize("stale", [None, b"x" * 128, _stale(999, 0xEE)], ids=["fresh", "old-format", "other-nonce"])
def test_two_ranks_exchange_an_id_and_ignore_a_stale_file(tmp_path, stale):
    # rank 1 starts first and finds whatever an earlier run left; rank 0 arrives 0.3 s later
    path, out = _run(tmp_path, stale)
    (rc0, b0), = out[0]
    (rc1, b1), = out[1]
    assert rc0 == 0 and rc1 == 0
    assert b1 == b0 == bytes(i & 255 for i in range(128))
    assert not os.path.exists(path)  # removed by rank 0 once the communicator would exist


def test_a_second_rendezvous_at_the_same_path_has_its_own_nonce(tmp_path):
    path, out = _run(tmp_path, None, rounds=2)
    for r in range(2):
        assert out[0][r][0] == 0 and out[1][r][0] == 0
        assert out[1][r][1] == out[0][r][1] == bytes((17 * r + i) & 255 for i in range(128))
    assert not os.path.exists(path)


def test_a_file_of_the_same_nonce_from_a_dead_run_is_replaced_not_trusted(tmp_path):
    # the worst case: a crashed run with the same nonce (a launcher that reuses its port and run id) left its file;
    # rank 0 removes it before publishing, so a rank that read the old bytes first is the only exposure -- here rank 0
    # is first, rank 1 arrives after and must see the new bytes
    path, out = _run(tmp_path, _stale(12345 * 1000003, 0xEE), delays=(0.0, 0.5))
    assert out[0][0][0] == 0 and out[1][0][0] == 0
    assert out[1][0][1] == bytes(i & 255 for i in range(128))


def test_no_rank_zero_is_an_error_not_a_hang(tmp_path):
    L = _lib()
    L.cpecan_em_set_rendezvous_nonce(7)
    buf = (C.c_ubyte * 128)()
    t0 = time.time()
    assert L.cpecan_em_rendezvous_exchange(str(tmp_path / "none.id").encode(), 1, 2, buf, 128, 300) != 0
    assert time.time() - t0 < 5 and b"no RCCL id" in L.cpecan_em_last_error()
