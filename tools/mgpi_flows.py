#!/usr/bin/env python3
"""The reference's train-once / query-many command-line flows on the GPU path, with `.npz` (or, where h5py exists, HDF)
containers that carry the reference's field names:

    mgpi_flows.py compress  data.ini interpolator.ini out.npz [--section NAME] [--kernel-params rows.csv]
    mgpi_flows.py predict   compressed.npz target_table interpolator.ini out.npz
    mgpi_flows.py condition data.ini target_table interpolator.ini out.npz [--diag]

mirrors bin/mgpi-compress:51-90, bin/mgpi-predict:58-94 and bin/mgpi-condition:56-84 (argument meaning, verbose lines,
the containers of mgpi/io.py:294-387).  x is uploaded once per flow and every parameter row reuses the resident buffers
(Interpolator.compress_many / predict_average).  One deliberate difference: mgpi-compress saves only the LAST parameter
row (its loop variable shadows the list, bin/mgpi-compress:67); here every row is saved, which is what mgpi-predict
expects when it zips `compressed` with `params`."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest='flow', required=True)
    c = sub.add_parser('compress')
    c.add_argument('data'); c.add_argument('interpolator'); c.add_argument('outpath')
    c.add_argument('--section', default=None); c.add_argument('--kernel-params', default=None)
    p = sub.add_parser('predict')
    p.add_argument('compressed'); p.add_argument('target'); p.add_argument('interpolator'); p.add_argument('outpath')
    d = sub.add_parser('condition')
    d.add_argument('data'); d.add_argument('target'); d.add_argument('interpolator'); d.add_argument('outpath')
    d.add_argument('--section', default=None); d.add_argument('--diag', action='store_true', help='pointwise variance instead of the full covariance')
    for s in (c, p, d):
        s.add_argument('-v', '--verbose', action='store_true')
    args = ap.parse_args(argv)
    interp = mgpi.parse_interpolator(args.interpolator, verbose=args.verbose)
    if args.flow == 'compress':
        (source_x, source_f), (xcols, fcol) = mgpi.parse_table(args.data, section=args.section, verbose=args.verbose)
        if args.kernel_params:
            table = mgpi.load_ascii_data(args.kernel_params, verbose=args.verbose)
            params = np.transpose([table[name] for name in table.dtype.names])
        else:
            params = np.array([interp.kernel.params], dtype=float)
        compressed = interp.compress_many(source_x, source_f, params, verbose=args.verbose)
        mgpi.save_compressed_data(args.outpath, source_x, compressed, params, xcols=xcols, fcol=fcol, verbose=args.verbose)
    elif args.flow == 'predict':
        source_x, compressed, params, xcols, fcol = mgpi.load_compressed_data(args.compressed, verbose=args.verbose)
        target_x = mgpi.load_target_data(args.target, xcols, verbose=args.verbose)
        mean = interp.predict_average(target_x, source_x, compressed, params, verbose=args.verbose)
        mgpi.save_conditioned_data(args.outpath, target_x, mean, xcols=xcols, fcol=fcol, verbose=args.verbose)
    else:
        (source_x, source_f), (xcols, fcol) = mgpi.parse_table(args.data, section=args.section, verbose=args.verbose)
        target_x = mgpi.load_target_data(args.target, xcols, verbose=args.verbose)
        if args.diag:
            mean, var = interp.condition_diag(target_x, source_x, source_f, verbose=args.verbose)
            mgpi.save_conditioned_data(args.outpath, target_x, mean, cov=var, xcols=xcols, fcol=fcol, verbose=args.verbose)
        else:
            mean, cov = interp.condition(target_x, source_x, source_f, verbose=args.verbose)
            mgpi.save_conditioned_data(args.outpath, target_x, mean, cov=cov, xcols=xcols, fcol=fcol, verbose=args.verbose)


if __name__ == '__main__':
    main()
