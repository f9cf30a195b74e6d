"""CPU oracle for the cpprob::inference(sis/smc) hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (cpprob_amd/) never does.  See cpprob_oracle.c for what is restated
(with reference file:line) and what is pinned against what.
"""
