"""Hazard check of the LDS ring schedule of the 256x128 two-workgroups-per-CU GEMM loop (csrc/gemm_core_pp2.h), on the host.
Units of a K step t: u = 3t + j, j = 0: A0 (rows 0..127), 1: B (128 columns), 2: A1 (rows 128..255); 16 KiB each, 4 LDS-DMA
instructions per thread (256 threads); ring of 5 slots, slot(u) = u % 5.  Phases p = 4t + q; reads (in the LOAD segment of the
phase, i.e. between barrier p-1 and barrier p): q=0: A0(t), B(t); q=1: B(t); q=2: A1(t); q=3: none.  Stages (also in the load
segment, before the phase's counted vmcnt wait and barrier): q=0: B(t+1), q=2: A1(t+1), q=3: A0(t+2); the prologue stages A0(0),
B(0), A1(0), A0(1), waits vmcnt(8) and takes a barrier.
Rules (MI355X guide + gemm_core_pp.h): RAW -- a unit read in phase p must be covered by every wave's vmcnt wait of a phase < p
(the waits sit in front of the barriers); WAR -- a slot may be re-staged in phase p only if its last read was in a phase <= p - 2
(one barrier per phase: the readers' lgkmcnt(0) of phase p-2 precedes their arrival at barrier p-1)."""
import sys


def simulate(nk):
    LOADS = 4
    issued = []                    # (unit, phase_issued) in issue order; prologue = phase -1
    first_read, last_read = {}, {}
    for t in range(nk):
        first_read[3 * t + 0], last_read[3 * t + 0] = 4 * t, 4 * t
        first_read[3 * t + 1], last_read[3 * t + 1] = 4 * t, 4 * t + 1
        first_read[3 * t + 2], last_read[3 * t + 2] = 4 * t + 2, 4 * t + 2
    nunits = 3 * nk
    covered = {}                   # unit -> phase whose wait covers it
    staged_phase = {}

    def stage(u, p):
        if u >= nunits:
            return False
        issued.append(u)
        staged_phase[u] = p
        if u >= 5:                 # WAR on the unit it overwrites
            assert last_read[u - 5] <= p - 2, ("WAR", nk, u, p, last_read[u - 5])
        return True

    def wait(p, vmcnt):            # vmcnt counts load INSTRUCTIONS allowed in flight (youngest first)
        units_in_flight = vmcnt // LOADS
        done = issued[:len(issued) - units_in_flight] if units_in_flight else issued[:]
        for u in done:
            covered.setdefault(u, p)

    for u in (0, 1, 2, 3):
        stage(u, -1)
    wait(-1, 8)
    counts = {}
    for t in range(nk):
        mode = 0 if t < nk - 2 else (1 if t == nk - 2 else 2)
        # q = 0
        s = stage(3 * (t + 1) + 1, 4 * t) if mode <= 1 else False
        wait(4 * t, 12 if mode <= 1 else 4)
        # q = 1
        wait(4 * t + 1, 8 if mode <= 1 else 0)
        # q = 2
        s = stage(3 * (t + 1) + 2, 4 * t + 2) if mode <= 1 else False
        wait(4 * t + 2, 12 if mode <= 1 else 0)
        # q = 3
        s = stage(3 * (t + 2) + 0, 4 * t + 3) if mode == 0 else False
        wait(4 * t + 3, 8 if mode == 0 else (4 if mode == 1 else 0))
    for u in range(nunits):
        assert u in staged_phase, ("never staged", nk, u)
        assert u in covered and covered[u] < first_read[u], ("RAW", nk, u, covered.get(u), first_read[u])
    return True


for nk in range(3, 70):
    simulate(nk)
print("pp2 schedule ok: RAW and WAR hold for 3..69 K steps")
