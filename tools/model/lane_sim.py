"""Cost model of wave scheduling policies for the transport kernel (round 2 design study).

A wave has 64 lanes; every context (photon) alternates between a walk of n voxel steps (n ~ geometric, mean 3:
measured 61 steps / 21 collisions per photon on les480) and an event.  One wave-iteration of the walk block costs
cA wave-instructions whatever the number of lanes walking, the event block cE, a context switch block cS.
Prints wave-instructions per collision cycle and lane utilisation for
  K=1       the round-1 structure (walk while >= thr lanes fly, then serve every waiting lane)
  private K every lane owns K contexts (registers + its own LDS slots), no cross-lane traffic
  pooled N  N contexts per wave in LDS, lanes take any ready context (compaction through wave-private lists)
"""
import numpy as np
rng = np.random.default_rng(1)

def steps(n, p=1/3.0):
    return rng.geometric(p, n)

def sim_private(K, cA=50, cE=280, cS=24, thrA=16, thrE=48, ncycle=200000):
    # state per lane/context: remaining steps (>0 walking, 0 needs event)
    rem = steps(64 * K).reshape(64, K)
    cur = np.zeros(64, int)            # context in registers
    instr = 0; coll = 0; useA = 0; useE = 0; nA = 0; nE = 0
    while coll < ncycle:
        # ---- walk phase
        while True:
            r = rem[np.arange(64), cur]
            fly = r > 0
            # lanes whose current context waits but which own a walking one would like to switch
            other = (rem > 0).any(axis=1) & ~fly
            if K > 1 and other.sum() >= 8:
                instr += cS
                for l in np.nonzero(other)[0]:
                    cur[l] = int(np.argmax(rem[l] > 0))
                continue
            if fly.sum() < thrA and ((rem == 0).any(axis=1)).sum() > 0:
                if K > 1 and other.sum() > 0 and fly.sum() + other.sum() >= thrA:
                    instr += cS
                    for l in np.nonzero(other)[0]:
                        cur[l] = int(np.argmax(rem[l] > 0))
                    continue
                break
            if fly.sum() == 0: break
            instr += cA; nA += 1; useA += fly.sum()
            rem[np.arange(64)[fly], cur[fly]] -= 1
        # ---- event phase: every lane with a waiting context serves one of them (repeat while enough lanes have one)
        while True:
            wait = (rem == 0)
            lanes = wait.any(axis=1)
            if lanes.sum() == 0: break
            if lanes.sum() < thrE and (rem > 0).any(): 
                if nE and lanes.sum() < thrE and ((rem > 0).any(axis=1)).sum() >= thrA: break
            instr += cE + (cS if K > 1 else 0); nE += 1; useE += lanes.sum()
            for l in np.nonzero(lanes)[0]:
                j = int(np.argmax(wait[l])); rem[l, j] = steps(1)[0]; cur[l] = j
            coll += lanes.sum()
    return instr / coll, useA / (64.0 * nA), useE / (64.0 * nE)

def sim_pooled(N, cA=50, cE=280, cS=30, thr_refill=40, ncycle=200000):
    rem = steps(N)
    inreg = np.arange(64)               # context index held by each lane (-1 none)
    instr = 0; coll = 0; useA = 0; useE = 0; nA = 0; nE = 0
    while coll < ncycle:
        # walk until the event list is long enough for a full event wave
        while True:
            have = inreg >= 0
            fly = np.zeros(64, bool); fly[have] = rem[inreg[have]] > 0
            nwait_total = (rem == 0).sum()
            if nwait_total >= 64 or (fly.sum() == 0 and not ((rem > 0).sum() > 0)): break
            if fly.sum() < thr_refill:
                # compaction: lanes whose context waits drop it and take a walking one from the pool
                instr += cS
                held = set(inreg[have & fly].tolist())
                free = [c for c in np.nonzero(rem > 0)[0] if c not in held]
                idle = np.nonzero(~fly)[0]
                inreg[idle] = -1
                for l, c in zip(idle, free): inreg[l] = c
                have = inreg >= 0
                fly = np.zeros(64, bool); fly[have] = rem[inreg[have]] > 0
                if fly.sum() == 0: break
            instr += cA; nA += 1; useA += fly.sum()
            rem[inreg[fly]] -= 1
        wait = np.nonzero(rem == 0)[0][:64]
        instr += cE + cS; nE += 1; useE += len(wait)
        rem[wait] = steps(len(wait)); coll += len(wait)
        # the event lanes now hold fresh walking contexts; contexts that were walking in registers were parked (cost in cS)
        inreg[:] = -1; inreg[:len(wait)] = wait
    return instr / coll, useA / (64.0 * nA), useE / (64.0 * nE)

if __name__ == '__main__':
    print('ideal (all lanes busy): %.1f wave-instr per 64 collisions -> per collision %.2f' % (3 * 50 + 280, (3 * 50 + 280) / 64.0))
    for K in (1, 2, 3):
        for thrA, thrE in ((16, 48), (24, 40), (32, 32), (8, 56)):
            c, ua, ue = sim_private(K, thrA=thrA, thrE=thrE, ncycle=60000)
            print('private K=%d thrA=%2d thrE=%2d : %.2f wave-instr per collision   walk util %.2f  event util %.2f' % (K, thrA, thrE, c, ua, ue))
    for N in (96, 128, 192):
        for thr in (32, 48):
            c, ua, ue = sim_pooled(N, thr_refill=thr, ncycle=60000)
            print('pooled  N=%3d refill<%2d     : %.2f wave-instr per collision   walk util %.2f  event util %.2f' % (N, thr, c, ua, ue))
