"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's descriptor-extraction path, used as the checker by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
egonn_amd/ may import from here; the product path fails loudly without its HIP library.
"""
