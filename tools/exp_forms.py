"""A/B of the rollout kernel forms (developer tool, GPU box): one lane per env / two lanes per env / 2 or 4 waves per
group of 64 envs, same seeds -> the records and the final state must be bit-identical, then microseconds per
256-ply launch (HIP events over `reps` launches after a warm-up at sustained clocks).
usage: python tools/exp_forms.py [board envs]..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-selfplay-mnk_amd")]
import torch
import mnk_hip
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.random_rollout import RandomRollout

DEV = "cuda:0"
FORMS = {"lane": ("0", None), "pair": ("1", "pair"), "pairw": ("1", "pairw"), "ws2": ("0", "ws2"), "ws4": ("0", "ws4")}


def set_form(name):
    pair, form = FORMS[name]
    os.environ["MNK_ROLLOUT_PAIR"] = pair
    if form:
        os.environ["MNK_ROLLOUT_FORM"] = form
    else:
        os.environ.pop("MNK_ROLLOUT_FORM", None)
    import mnk_hip

    mnk_hip.reload_config()  # the library reads its environment knobs once


def run(board, nenv, chunk=256, reps=40, record=True, log=False):
    m, n, k = (int(v) for v in board.split("x"))
    ref = None
    row = []
    for name in FORMS:
        set_form(name)
        env = TorchVectorMnkEnv(m, n, k, nenv, device=DEV)
        roll = RandomRollout(env, seed=7)
        buf = roll.alloc(chunk, log_actions=log) if record else None
        roll.run(chunk, out=buf, record=record)
        roll.run(chunk, out=buf, record=record)
        sig = (env._planes.clone(), env._meta.clone(), buf.planes.clone() if record else None,
               buf.meta.clone() if record else None, roll.stats.clone(), buf.act.clone() if log else None)
        if ref is None:
            ref = sig
        else:
            for a, b in zip(ref, sig):
                assert (a is None and b is None) or torch.equal(a, b), f"{board} {name}: differs from the one-lane form"
        for _ in range(150):  # sustained clocks
            roll.run(chunk, out=buf, record=record)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            roll.run(chunk, out=buf, record=record)
        e1.record()
        torch.cuda.synchronize()
        row.append((name, e0.elapsed_time(e1) * 1e3 / reps))
    rows = mnk_hip.record_words(m, n)
    nbytes = nenv * (chunk * (8 * rows + 4) + 2 * (16 * mnk_hip.state_words(m, n) + 4)) if record else 0
    txt = "  ".join(f"{name} {us:7.1f} us" + (f" ({nbytes / us / 1e3:5.0f} GB/s)" if record else "") for name, us in row)
    print(f"{board} x {nenv:6d} rec={int(record)} log={int(log)}: {txt}", flush=True)


if __name__ == "__main__":
    mnk_hip.load()
    args = sys.argv[1:]
    cases = [(args[i], int(args[i + 1])) for i in range(0, len(args), 2)] or [
        ("9x9x5", 65536), ("9x9x5", 32768), ("9x9x5", 16384), ("9x9x5", 131072),
        ("19x19x5", 32768), ("19x19x5", 16384), ("19x19x5", 65536)]
    for board, nenv in cases:
        run(board, nenv)
    run("9x9x5", 65536, record=False)
    run("19x19x5", 32768, record=False)
    run("9x9x5", 65536, log=True)
    run("19x19x5", 32768, log=True)
