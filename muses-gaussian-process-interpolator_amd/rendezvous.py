"""Getting the ranks of one node together: who am I, and how does rank 0's RCCL unique id reach the others.

One process per GPU, started either by an external launcher that exports RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_PORT (only those environment variables are used, nothing of the launcher is imported) or by ``spawn_ranks``
below.  The 128-byte unique id travels through a file (one per exchange, numbered): the ranks of one node share a
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


_EXCHANGES = [0]     # how many ids this process has exchanged: every rank of a job counts the same calls


def exchange_unique_id(rank, world, make_id, timeout=600.0):
    """rank 0 calls make_id() (-> bytes) and publishes it; every rank returns the same bytes.
    Every call of a job has its own generation: the file of call number s is ``<path>.<s>``, so a second communicator in the
    same job (a second transport, a reused MGPI_RENDEZVOUS) can never pick up the id of an earlier one - mismatched ids make
    ncclCommInitRank hang, which no timeout here could cover.  The file is created exclusively with mode 0600."""
    if world == 1:
        return make_id()
    seq = _EXCHANGES[0]
    _EXCHANGES[0] += 1
    path = '%s.%d' % (_id_path(), seq)
    if rank == 0:
        blob = make_id()
        tmp = '%s.%d.tmp' % (path, os.getpid())
        if os.path.exists(tmp):
            os.remove(tmp)
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, 'wb') as fh:
            fh.write(b'%d %d\n' % (os.getpid(), seq) + blob)
        os.replace(tmp, path)                      # atomic: a reader sees nothing or everything
        import atexit
        atexit.register(lambda: os.path.exists(path) and os.remove(path))
        return blob
    deadline = time.time() + timeout
    while time.time() < deadline:
        try:
            with open(path, 'rb') as fh:
                head, blob = fh.read().split(b'\n', 1)
            pid, got = (int(v) for v in head.split())
            if got == seq and (os.environ.get('MGPI_RENDEZVOUS') or _sibling_alive(pid)):
                return blob
        except (OSError, ValueError):
            pass
        time.sleep(0.02)
    raise RuntimeError('rank %d: no unique id (generation %d) from rank 0 within %.0f s (%s)' % (rank, seq, timeout, path))


def _tail(path, limit=4000):
    try:
        with open(path, 'rb') as fh:
            fh.seek(0, os.SEEK_END)
            size = fh.tell()
            fh.seek(max(0, size - limit))
            return fh.read().decode('utf-8', 'replace')
    except OSError:
        return ''


def spawn_ranks(argv, world, env=None, timeout=1800.0, report=None):
    """start `world` copies of ``python argv...`` with RANK / LOCAL_RANK / WORLD_SIZE / MGPI_RENDEZVOUS set and watch them
    ALL: the first rank that exits non-zero (or the deadline) ends the job - the others are killed instead of being left
    blocked in a collective that has lost its peer - and the stderr tail of every rank is written to `report` (default:
    this process's stderr).  Returns the list of exit codes (-9 for a rank this function killed, None never).
    Rank 0 inherits stdout; every rank's stderr goes to a file of its own.  The caller must not have touched the GPU: the
    children are new processes created by fork + exec, and killing them from here is safe for the same reason."""
    base = dict(os.environ if env is None else env)
    report = sys.stderr if report is None else report
    workdir = tempfile.mkdtemp(prefix='mgpi_ranks_')
    path = os.path.join(workdir, 'rdzv')
    procs, logs = [], []
    for r in range(world):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MGPI_RENDEZVOUS=path)
        e.setdefault('MASTER_ADDR', '127.0.0.1')
        log = open(os.path.join(workdir, 'rank%d.stderr' % r), 'wb')
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e, stdout=None if r == 0 else subprocess.DEVNULL,
                                      stderr=log))
    deadline = None if timeout is None else time.time() + timeout
    why = None
    try:
        while True:
            codes = [pr.poll() for pr in procs]
            bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                why = 'rank %d exited with code %d' % (bad[0], codes[bad[0]])
                break
            if all(c == 0 for c in codes):
                break
            if deadline is not None and time.time() > deadline:
                why = 'no result within %.0f s' % timeout
                break
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        codes = [pr.wait() for pr in procs]
        for log in logs:
            log.close()
        if why is not None:
            report.write('spawn_ranks: %s; the other ranks were stopped.  Exit codes: %s\n' % (why, codes))
            for r in range(world):
                tail = _tail(os.path.join(workdir, 'rank%d.stderr' % r))
                if tail.strip():
                    report.write('---- rank %d stderr (tail) ----\n%s\n' % (r, tail))
        else:
            tail0 = _tail(os.path.join(workdir, 'rank0.stderr'))
            if tail0.strip():
                report.write(tail0)
        import shutil
        shutil.rmtree(workdir, ignore_errors=True)
    if why is not None and all(c == 0 for c in codes):
        codes[0] = 124                       # every rank was still fine when the deadline passed
    return codes
