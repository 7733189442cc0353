"""TEST INFRASTRUCTURE ONLY - never imported by the product (`care_amd/`).

`oracle/` holds the CPU restatement of the reference's captioning forward path
(`care_cpu.py`), the script that imports the genuine reference from /root/reference
in the build container to pin that restatement (`gen_golden.py` -> tests/golden/),
and nothing else.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import from here, and only as the checker / the timed baseline.

Parity status: PINNED against outputs of the reference itself run in the build
container (the reference ships no tests or golden vectors of its own).
There is no C restatement: the path is floating-point tensor math whose reference
arithmetic *is* torch CPU kernels, so the restatement is torch-fp32 on CPU.
"""
