"""Getting the ranks of one node together: who am I, and how does rank 0's RCCL unique id reach the others.

One process per GPU, started either by an external launcher that exports RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_PORT (only those environment variables are used, nothing of the launcher is imported) or by ``spawn_ranks``
below.  The 128-byte unique id travels through a file: the ranks of one node share a
filesystem, and nothing else has to be agreed on.  (The reference has no multi-process mode at all - SURVEY.md
section 5.)
"""
import os
import subprocess
import sys
import tempfile
import time

__all__ = ['rank_env', 'exchange_unique_id', 'spawn_ranks']


def rank_env():
    """(rank, local_rank, world) from the launcher's environment; (0, 0, 1) when there is none"""
    rank = int(os.environ.get('RANK', '0'))
    return rank, int(os.environ.get('LOCAL_RANK', str(rank))), int(os.environ.get('WORLD_SIZE', '1'))


def _id_path():
    """one name per job: MGPI_RENDEZVOUS if the launcher set it (spawn_ranks does), else derived from what the ranks
    of one job share - their parent process (the launcher) and its master port"""
    explicit = os.environ.get('MGPI_RENDEZVOUS')
    if explicit:
        return explicit
    return os.path.join(tempfile.gettempdir(), 'mgpi_rdzv_%s_%d' % (os.environ.get('MASTER_PORT', '0'), os.getppid()))


def _sibling_alive(pid):
    """is `pid` a live process with the same parent as this one (i.e. rank 0 of THIS job, not a stale file's author)"""
    try:
        with open('/proc/%d/stat' % pid) as fh:
            fields = fh.read().rsplit(')', 1)[1].split()
        return int(fields[1]) == os.getppid()
    except (OSError, ValueError, IndexError):
        return False


def exchange_unique_id(rank, world, make_id, timeout=600.0):
    """rank 0 calls make_id() (-> bytes) and publishes it; every rank returns the same bytes"""
    if world == 1:
        return make_id()
    path = _id_path()
    if rank == 0:
        blob = make_id()
        tmp = '%s.%d.tmp' % (path, os.getpid())
        with open(tmp, 'wb') as fh:
            fh.write(b'%d\n' % os.getpid() + blob)
        os.replace(tmp, path)                      # atomic: a reader sees nothing or everything
        if not os.environ.get('MGPI_RENDEZVOUS'):      # spawn_ranks removes its own file; here nobody else would
            import atexit
            atexit.register(lambda: os.path.exists(path) and os.remove(path))
        return blob
    deadline = time.time() + timeout
    while time.time() < deadline:
        try:
            with open(path, 'rb') as fh:
                head, blob = fh.read().split(b'\n', 1)
            if os.environ.get('MGPI_RENDEZVOUS') or _sibling_alive(int(head)):
                return blob
        except (OSError, ValueError):
            pass
        time.sleep(0.02)
    raise RuntimeError('rank %d: no unique id from rank 0 within %.0f s (%s)' % (rank, timeout, path))


def spawn_ranks(argv, world, env=None, timeout=None):
    """start `world` copies of ``python argv...`` with RANK / LOCAL_RANK / WORLD_SIZE / MGPI_RENDEZVOUS set and wait for
    them; returns the list of exit codes.  Rank 0 inherits stdout.  The caller must not have touched the GPU: the
    children are new processes created by fork + exec."""
    base = dict(os.environ if env is None else env)
    fd, path = tempfile.mkstemp(prefix='mgpi_rdzv_')
    os.close(fd)
    os.remove(path)
    procs = []
    for r in range(world):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MGPI_RENDEZVOUS=path)
        e.setdefault('MASTER_ADDR', '127.0.0.1')
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e, stdout=None if r == 0 else subprocess.DEVNULL))
    codes = []
    deadline = None if timeout is None else time.time() + timeout
    try:
        for pr in procs:
            left = None if deadline is None else max(1.0, deadline - time.time())
            codes.append(pr.wait(timeout=left))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        if os.path.exists(path):
            os.remove(path)
    return codes
