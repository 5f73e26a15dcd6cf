"""oracle/ -- TEST INFRASTRUCTURE ONLY (never imported by pycllp_amd/).

Two checkers live here:

* ``oracle.port``    -- our own CPU restatement (C, ``ipm_dense_ref.c`` + a small numpy twin) of the
  reference's batched dense primal-normal interior-point path (pycllp/cl/primal_normal.cl,
  pycllp/cl/ldl.cl).  Pinned against the reference by tests/golden/*.npz.
* ``oracle.hsd_ref`` -- ctypes binding of ``oracle/_ref/libhsd_ref.so``: the reference's OWN CPU solver
  (pycllp/ipo.py -> pycllp/ipo/hsd.c) compiled from the reference sources by ``oracle/Makefile``.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
