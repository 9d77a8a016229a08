"""TEST INFRASTRUCTURE ONLY — CPU oracle (oracle.oracle) and the host build of the reference's own kernels (oracle.ref).
Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from jnerf_amd/."""
