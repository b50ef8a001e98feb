"""small host utilities the hot path's callers rely on (mgpi/utils.py)"""
import numpy as np

__all__ = ['seed', 'factory', 'construct_logprior', 'subdivide_data']


def seed(s, verbose=False):
    """set numpy's random seed (mgpi/utils.py:11-16, whose body references an undefined ``args`` and ``np.seed`` and
    cannot run; this is what it says it does)"""
    if verbose:
        print('setting numpy.seed=%d' % s)
    np.random.seed(s)


def factory(klass):
    """{class name: class} for klass and everything derived from it - how the reference's INI loader finds
    kernels and interpolators by name (mgpi/utils.py:20-26, used at mgpi/io.py:408,537)"""
    found = {klass.__name__: klass}
    pending = list(klass.__subclasses__())
    while pending:
        sub = pending.pop()
        if sub.__name__ not in found:
            found[sub.__name__] = sub
            pending.extend(sub.__subclasses__())
    return found


def construct_logprior(names, bounds, fixed):
    """box prior over the FREE parameters (those of `names` not in `fixed`): 0 inside every (min, max) of `bounds`,
    -inf outside (mgpi/utils.py:30-48); the function takes the free-parameter vector in `names` order"""
    free = [name for name in names if name not in fixed]
    for key in bounds.keys():
        assert key in free, 'cannot set prior for unknown parameter: ' + key
    by_index = dict((free.index(key), val) for key, val in bounds.items())

    def logprob(params):
        for ind, (lo, hi) in by_index.items():
            if (params[ind] < lo) or (hi < params[ind]):
                return -np.inf
        return 0.0

    return logprob


def subdivide_data(source_x, source_f, frac=0.5, verbose=False):
    """random split into (train_x, train_f), (test_x, test_f) with a fraction `frac` in the training set
    (mgpi/utils.py:52-73; numpy's global random state, like upstream)"""
    num = len(source_x)
    num_train = int(round(frac * num, 0))
    if verbose:
        print('dividing data set of %d samples into %d training samples and %d testing samples' % (num, num_train, num - num_train))
    assert num_train > 0, 'will not partion data into training set with zero samples'
    assert num > num_train, 'will not partion data into training set with zero samples'
    inds = np.arange(num)
    np.random.shuffle(inds)
    return (source_x[inds[:num_train]], source_f[inds[:num_train]]), (source_x[inds[num_train:]], source_f[inds[num_train:]])
