"""A small host-side process group over TCP (127.0.0.1): the channel that carries the 128-byte RCCL id of a stream
group, the barrier and the max-over-ranks time of bench.py - without torch.  One process per GPU of ONE node, started
by any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK (`python -m torch.distributed.run` does; only its launcher is
used, no `import torch` happens in the ranks).

Rank 0 listens on an ephemeral port and publishes it in a file that every rank of the job can name: the directory is
$TMPDIR (default /tmp), the name is built from MASTER_PORT, TORCHELASTIC_RUN_ID and TORCHELASTIC_RESTART_COUNT (without
a MASTER_PORT: from the launcher's pid, the parent of every rank), or given by CHRONOCLUST_RDZV_FILE.  The launcher's
own port (MASTER_PORT) is left alone: with `torch.distributed.run` the agent's store is bound to it.

The operations are what the harness needs, star-shaped through rank 0, payloads are small:
    all_gather_bytes(b)  -> [bytes of rank 0, ..., bytes of rank W-1] on every rank
    barrier(), broadcast_bytes(b, src), all_equal(b), max_float(x)
and the handful of `torch.distributed`-style methods chronoclust_amd.multi calls (get_rank, get_world_size,
is_initialized, all_gather_object, broadcast_object_list), so the same helpers serve either kind of group.
Every wait has a deadline (default 300 s): a rank that is gone raises TimeoutError / ConnectionError in the others
instead of hanging them."""
import json
import os
import socket
import struct
import tempfile
import time

_MAGIC = b"CCRDZV1\n"


class HostGroup(object):
    def __init__(self, rank, world, rdzv_file=None, timeout=300.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self._file = rdzv_file or default_rdzv_file()
        self._peers = {}      # rank 0: rank -> socket
        self._sock = None     # other ranks: the connection to rank 0
        self._listener = None
        if self.world > 1:
            (self._serve if self.rank == 0 else self._join)()

    # ---- set-up -------------------------------------------------------------------------------------------

    def _serve(self):
        ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        ls.bind(("127.0.0.1", 0))
        ls.listen(self.world)
        ls.settimeout(self.timeout)
        self._listener = ls
        self._token = os.urandom(8).hex()
        # the port and the token that admits a rank: readable by this user only, written to a name nobody can predict
        # (mkstemp: O_CREAT | O_EXCL, mode 0600, so no symlink can be planted under it) and moved into place
        fd, tmp = tempfile.mkstemp(prefix=os.path.basename(self._file) + ".", suffix=".tmp",
                                   dir=os.path.dirname(self._file) or ".")
        try:
            with os.fdopen(fd, "w") as f:
                f.write("%d %s\n" % (ls.getsockname()[1], self._token))
            os.replace(tmp, self._file)  # (a stale file of an earlier job with the same name is overwritten)
        except BaseException:
            try:
                os.unlink(tmp)
            except OSError:
                pass
            raise
        deadline = time.monotonic() + self.timeout
        while len(self._peers) < self.world - 1:
            if time.monotonic() > deadline:
                raise TimeoutError("rendezvous: %d of %d ranks joined within %.0f s" % (
                    len(self._peers) + 1, self.world, self.timeout))
            try:
                c, _ = ls.accept()
            except socket.timeout:
                continue
            c.settimeout(self.timeout)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            try:
                hello = _recv_exact(c, len(_MAGIC) + 16 + 4)
            except (ConnectionError, socket.timeout):
                c.close()
                continue
            r = struct.unpack("<i", hello[-4:])[0]
            if hello[:len(_MAGIC)] != _MAGIC or hello[len(_MAGIC):-4] != self._token.encode() or not (0 < r < self.world) \
                    or r in self._peers:
                c.close()  # not a rank of this job (a stale file pointed it here)
                continue
            c.sendall(b"OK")
            self._peers[r] = c

    def _join(self):
        deadline = time.monotonic() + self.timeout
        last = None
        while time.monotonic() < deadline:
            try:
                with open(self._file) as f:
                    port_s, token = f.read().split()
                c = socket.create_connection(("127.0.0.1", int(port_s)), timeout=5.0)
                c.settimeout(self.timeout)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.sendall(_MAGIC + token.encode() + struct.pack("<i", self.rank))
                if _recv_exact(c, 2) == b"OK":
                    self._sock = c
                    return
                c.close()
            except (OSError, ValueError) as e:  # file not there yet / stale file / rank 0 not listening yet
                last = e
            time.sleep(0.05)
        raise TimeoutError("rendezvous: rank %d could not reach rank 0 through %s within %.0f s (%s)" % (
            self.rank, self._file, self.timeout, last))

    # ---- collectives --------------------------------------------------------------------------------------

    def all_gather_bytes(self, payload):
        payload = bytes(payload)
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [None] * (self.world - 1)
            for r, c in self._peers.items():
                parts[r] = _recv_msg(c)
            blob = _pack_parts(parts)
            for c in self._peers.values():
                _send_msg(c, blob)
            return parts
        _send_msg(self._sock, payload)
        return _unpack_parts(_recv_msg(self._sock), self.world)

    def barrier(self):
        self.all_gather_bytes(b"")

    def broadcast_bytes(self, payload, src=0):
        return self.all_gather_bytes(payload if self.rank == src else b"")[src]

    def all_equal(self, payload):
        parts = self.all_gather_bytes(payload if isinstance(payload, bytes) else str(payload).encode())
        return all(p == parts[0] for p in parts)

    def max_float(self, x):
        return max(struct.unpack("<d", p)[0] for p in self.all_gather_bytes(struct.pack("<d", float(x))))

    # ---- the subset of torch.distributed that chronoclust_amd.multi uses ---------------------------------------

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.world

    def is_initialized(self):
        return True

    # (payloads are bytes or JSON-able values - what multi.py sends: the RCCL id, digests, small lists; nothing a peer
    # sends is ever unpickled or evaluated)
    def all_gather_object(self, box, obj):
        for i, p in enumerate(self.all_gather_bytes(_encode_obj(obj))):
            box[i] = _decode_obj(p)

    def broadcast_object_list(self, box, src=0):
        blob = _pack_parts([_encode_obj(x) for x in box]) if self.rank == src else b""
        got = self.broadcast_bytes(blob, src)
        box[:] = [_decode_obj(p) for p in _unpack_parts(got, len(box))]

    def close(self):
        for c in list(self._peers.values()) + [self._sock, self._listener]:
            if c is not None:
                try:
                    c.close()
                except OSError:
                    pass
        self._peers, self._sock, self._listener = {}, None, None
        if self.rank == 0 and self.world > 1:
            try:
                os.unlink(self._file)
            except OSError:
                pass


def default_rdzv_file():
    explicit = os.environ.get("CHRONOCLUST_RDZV_FILE")
    if explicit:
        return explicit
    port = os.environ.get("MASTER_PORT")
    # two jobs of one user at once differ in MASTER_PORT (a launcher cannot bind the same one twice); only a job started
    # without one falls back on the launcher's pid, which ranks behind a wrapper script would not share
    key = "_".join(str(x) for x in (port or "0", os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                    os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), "p" if port else os.getppid()))
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), "chronoclust_rdzv_%s_%d" % (key, os.getuid()))


def from_env(timeout=300.0):
    """The group of this job's ranks (RANK / WORLD_SIZE from the launcher; one process alone is a group of one)."""
    return HostGroup(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), timeout=timeout)


def _recv_exact(c, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = c.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous: a peer closed the connection")
        buf += chunk
    return bytes(buf)


def _send_msg(c, payload):
    c.sendall(struct.pack("<q", len(payload)) + payload)


def _recv_msg(c):
    return _recv_exact(c, struct.unpack("<q", _recv_exact(c, 8))[0])


def _pack_parts(parts):
    """[bytes, ...] -> one frame: count, then length-prefixed items."""
    return struct.pack("<i", len(parts)) + b"".join(struct.pack("<q", len(p)) + p for p in parts)


def _unpack_parts(blob, expect):
    n = struct.unpack_from("<i", blob, 0)[0]
    if n != expect:
        raise ConnectionError("rendezvous: malformed frame (%d parts, %d expected)" % (n, expect))
    out, off = [], 4
    for _ in range(n):
        ln = struct.unpack_from("<q", blob, off)[0]
        off += 8
        if ln < 0 or off + ln > len(blob):
            raise ConnectionError("rendezvous: malformed frame")
        out.append(bytes(blob[off:off + ln]))
        off += ln
    return out


def _encode_obj(obj):
    """bytes as they are; anything else as JSON (numbers, strings, lists, dicts, None): data, never code."""
    if isinstance(obj, (bytes, bytearray)):
        return b"B" + bytes(obj)
    return b"J" + json.dumps(obj).encode("utf-8")


def _decode_obj(p):
    tag, body = p[:1], p[1:]
    if tag == b"B":
        return body
    if tag == b"J":
        return json.loads(body.decode("utf-8"))
    raise ConnectionError("rendezvous: malformed object frame")
