"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT.

CPU restatement (fp32 PyTorch functions over a reference-compatible state_dict, plus plain C for the
assignment solver) of the TOIST/MDETR hot path.  It is the checker the HIP kernels are compared with
and the thing timed as `cpu_baseline` in bench.py.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; toist_amd/ never does.

Pinning: every function here is checked against golden vectors produced by importing the real
reference modules from /root/reference in the build container (tests/golden/make_golden.py wrote the
fixtures under tests/golden/); the solver is checked against SciPy (tests/test_oracle_lsap.py).
Unpinned pieces are listed in DESIGN.md: the ResNet-101 body (torchvision absent, so the restatement
follows the public v1.5 bottleneck architecture and is checked by parameter-count / shape
invariants only) and the RoBERTa tokenizer (no vocabulary files offline).
"""
