"""TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.

CPU oracle for the LarvaNet hot path. Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; larvanet_amd/ never does.

  larva_ref.c / larva_ref.py   plain-C restatement of each arithmetic step (+ numpy glue that
                               wires the steps into the reference's network graph)
  larva_torch.py               the same graph written with torch CPU operators (the operators
                               the reference itself calls), with autograd; this is the "reference
                               CPU path" that bench.py times as cpu_baseline (kind "port")

Parity status: PINNED by tests/golden/*.npz, generated in the build container by importing the
reference (tests/golden/make_golden.py). The reference ships no tests or golden vectors of its own.
"""
