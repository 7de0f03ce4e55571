"""What an invocation of bench.py will run: its phases in order, each with its watchdog budget (`bench.py --dry-run` prints it)."""
from . import gather
from .workloads import WORKLOADS

SECONDARY = ("xworld7_ego3", "simple_game", "simple_race")


def plan(args, world):
    """the phases this invocation would run, in order, with the watchdog budget (seconds; None = unguarded) of each"""
    guarded = world > 1 or args.force_exchange
    PT = args.phase_timeout if guarded else None
    is_xw = WORKLOADS[args.workload][0] == "xworld"
    fused = args.fused if not is_xw else 1
    ph = [{"phase": "main", "what": "set-up, frame gate, warm-up, spin, settle, %d + %d timed regions of %d steps, write ceiling"
           % (args.repeats, args.repeats, args.steps), "budget_s": 3 * PT if PT else None}]
    if fused == 1 and not args.autoreset:
        ph.append({"phase": "step_autoreset", "what": "3 regions of the fused call", "budget_s": PT})
    if is_xw and world == 1 and fused == 1 and not args.autoreset:
        ph.append({"phase": "classic_path", "what": "3 regions of a second batch on the classic kernel sequence", "budget_s": None})
    if (world == 1 and not args.force_exchange and not args.no_secondary and args.workload == "xworld7" and fused == 1
            and not args.envs_per_gpu and not args.autoreset):
        ph += [{"phase": "secondary:" + w, "what": "3 regions + parity slab", "budget_s": None} for w in SECONDARY]
    if guarded and not args.no_screens_gather:
        grids_ok = is_xw and not WORKLOADS[args.workload][1].get("visible_radius")
        ph += [{"phase": "screens_gather:" + m, "what": "%d regions" % args.repeats, "budget_s": PT} for m in gather.modes_for(args.gather, grids_ok)]
    if world > 1 and (args.c5 or world == 8) and args.workload != "xworld11":
        ph.append({"phase": "c5", "what": "xworld11: 3 regions device-resident + 3 per gather mode", "budget_s": 2 * PT if PT else None})
    ph.append({"phase": "parity", "what": "oracle rollout of %d envs over every recorded step" % args.parity_envs, "budget_s": None})
    if not args.no_cpu_baseline and world == 1:
        ph.append({"phase": "cpu_baseline", "what": "~10 s of the oracle on every host core", "budget_s": None})
    if guarded:
        ph.append({"phase": "teardown", "what": "barrier + destroy_process_group (the line is out)", "budget_s": PT})
    return ph
