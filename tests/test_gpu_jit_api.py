"""GPU: the API-level kernels specialised at run time (ABI 6, csrc/mnk_jit.hip).

The reference takes any board (env/torch_vector_mnk_env.py:9); the kernels behind ``env.step`` / ``observe`` /
``wrapper.step`` ... have compile-time geometry for five boards only, every other board used to stay on generic code for
good.  Now hiprtc instantiates the same kernel templates with the board's own geometry once a kernel is hot (or at the
first launch with ``MNK_JIT_API=1``), which also moves the board to the packed write-out and folds the masked draw into
the self-play step kernels for any row width.  The results must not change: the differential fuzzers of
test_gpu_fuzz.py and the folded-draw parity test of test_gpu_fused_draw.py run here with every API kernel specialised
from its first launch, on boards with 1 ... 21 words per plane, rows wider than 31 cells (table write-out on compile-time
geometry) and non-square shapes; then the switch itself: generic -> specialised in the middle of a game, nothing
compiled under a hipGraph capture, ``jit_prepare`` before one."""
import os

import numpy as np
import pytest
import torch

import test_gpu_fused_draw as fd
import test_gpu_fuzz as fz
import test_gpu_selfplay as sp
import test_gpu_sink as sk
from oracle.env_torch import OracleVectorEnv
from oracle.policies import LowestLegalPolicy

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

hip = fz.hip  # the module-scoped fixture of the fuzzers (env, wrapper, rollout, binding)


@pytest.fixture()
def jit_api(hip):
    """every API kernel of a board without a built-in variant is compiled at its first launch"""
    saved = {k: os.environ.get(k) for k in ("MNK_JIT_API", "MNK_JIT")}
    os.environ["MNK_JIT_API"] = "1"
    hip.lib.reload_config()
    yield hip.lib
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    hip.lib.reload_config()


# (m, n, k): words per plane 5, 2, 1, 5, 1, 11 (rows of 33 cells: no packed write-out), 3, 21 (625 cells)
SHAPES = [(12, 12, 5), (6, 7, 4), (5, 5, 4), (11, 11, 5), (4, 6, 3), (10, 33, 5), (7, 9, 7), (25, 25, 5)]


@pytest.mark.parametrize("shape", SHAPES)
def test_fuzzers_on_the_specialised_api_kernels(hip, jit_api, shape, monkeypatch):
    """every env operation, the wrapper (scripted opponents: pre + post; a third of its steps through the folded draw),
    the one-launch step: == the oracle after every operation, with the board's own kernels from the first launch on"""
    lib = jit_api
    monkeypatch.setattr(fz, "_shape", lambda rng: shape)
    fz._MAX_WRAPPER_STEPS[0] = 120
    try:
        base = 700 + 10 * SHAPES.index(shape)
        fz.test_env_fuzz(hip, base)
        fz.test_env_fuzz(hip, base + 1)
        fz.test_wrapper_fuzz(hip, base)       # LowestLegal opponent, f32 observations
        fz.test_wrapper_fuzz(hip, base + 1)   # HighestLegal, bf16
        fz.test_rollout_and_log_fuzz(hip, base)  # (its one-launch plies: mnk_step_random)
    finally:
        fz._MAX_WRAPPER_STEPS[0] = 10 ** 9
    m, n, k = shape
    for kind in (lib.JIT_API_STEP, lib.JIT_API_STEP_DRAW, lib.JIT_API_STEP_SUBSET, lib.JIT_API_OBSERVE, lib.JIT_API_SP_PRE,
                 lib.JIT_API_SP_POST):
        if kind == lib.JIT_API_OBSERVE and shape == (7, 9, 7):
            continue  # a kernel that never looks at k: this board shares the built-in 9-wide, 3-word variant (9x9x5's)
        assert lib.jit_api_ready(m, n, k, kind), (shape, kind, lib.load().mnk_jit_last_error())
    assert not lib.jit_api_ready(9, 9, 5, lib.JIT_API_STEP)  # a board with a built-in variant never gets one


@pytest.mark.parametrize("opponent", ["random", "scripted", "net"])
@pytest.mark.parametrize("m,n,k,nenv", [(12, 12, 5, 300), (6, 7, 4, 257), (5, 5, 4, 130), (10, 33, 5, 70), (11, 11, 5, 65536)])
def test_folded_draw_of_any_row_width(hip, jit_api, m, n, k, nenv, opponent):
    """``wrapper.step_logits`` on a board without a compile-time draw shape used to be two launches inside the call; its
    specialised kernels draw inside the step (mnk_draw::Shape by row width: the sampler's own shape, so the same action
    for the same uniform): == ``sampler.draw`` + ``wrapper.step``, f32 / bf16 / absent logits, stochastic and
    deterministic, every output and the whole state"""
    lib = jit_api
    if nenv == 65536 and opponent == "net":
        pytest.skip("the full-size batch runs with the random and the scripted opponent (a conv net at this batch is 20 s of MIOpen)")
    from alg.rollout_buffer import RolloutBuffer
    from selfplay import graphed, policy

    class NS:  # the namespace the fused-draw tests take from their own fixture
        pass

    ns = NS()
    ns.lib, ns.Env, ns.Wrapper, ns.policy, ns.graphed, ns.Buffer = lib, hip.Env, hip.Wrapper, policy, graphed, RolloutBuffer
    fd.test_step_logits_equals_sample_then_step(ns, m, n, k, nenv, opponent)
    which = 2 if opponent == "random" else 0
    for dtype in (torch.float32, torch.bfloat16, None):
        assert lib.jit_api_ready(m, n, k, lib.jit_api_draw_kind(which, dtype)), (m, n, k, dtype)
    if opponent == "net":
        assert lib.jit_api_ready(m, n, k, lib.jit_api_draw_kind(1, torch.float32))  # the opponent's draw inside `post`


def test_a_launch_that_would_not_fit_the_lds_stays_on_two_launches(hip, jit_api):
    """31x31: the packed write-out stage (61 KB) and the draw's slab of rows (31 KB) do not fit the 64 KB a launch may ask
    for together, so ``step_logits`` keeps its two launches there (sampler + the specialised plain kernel) -- decided
    on the host before anything is compiled, same results"""
    lib = jit_api
    from alg.rollout_buffer import RolloutBuffer
    from selfplay import graphed, policy

    class NS:
        pass

    ns = NS()
    ns.lib, ns.Env, ns.Wrapper, ns.policy, ns.graphed, ns.Buffer = lib, hip.Env, hip.Wrapper, policy, graphed, RolloutBuffer
    fd.test_step_logits_equals_sample_then_step(ns, 31, 31, 6, 70, "scripted")
    assert lib.jit_api_ready(31, 31, 6, lib.JIT_API_SP_PRE) and lib.jit_api_ready(31, 31, 6, lib.JIT_API_SP_POST)
    for dtype in (torch.float32, torch.bfloat16, None):
        assert not lib.jit_api_ready(31, 31, 6, lib.jit_api_draw_kind(0, dtype))


def test_reference_fixtures_through_the_specialised_kernels(hip, jit_api, golden_dir):
    """What the REFERENCE recorded on boards without a built-in variant (tests/golden, made by the imported reference:
    env op-logs of 4x6x3 and 7x9x7, the 4x6x3 wrapper trace, the edge scenarios on such boards), replayed on the
    board's own run-time compiled kernels: the fixtures pin the oracle, this pins the specialised kernels to the same
    data directly."""
    lib = jit_api
    from replay import golden_files, play_scenario, replay_env_log, replay_selfplay_trace
    from scenarios import SCENARIOS

    boards = set()
    for path in golden_files(golden_dir, "env_"):
        log = np.load(path)
        m, n, k, nenv, _ = (int(v) for v in log["geom"])
        if (m, n, k) in ((4, 6, 3), (7, 9, 7)):
            replay_env_log(hip.Env(m, n, k, nenv, device=DEV), log)
            replay_env_log(hip.Env(m, n, k, nenv, device=DEV, obs_dtype=torch.uint8), log)
            boards.add((m, n, k))
    assert boards == {(4, 6, 3), (7, 9, 7)}
    for (m, n, k) in boards:
        assert lib.jit_api_ready(m, n, k, lib.JIT_API_STEP) and lib.jit_api_ready(m, n, k, lib.JIT_API_STEP_SUBSET)
    traces = [p for p in golden_files(golden_dir, "selfplay_") if "4x6x3" in p]
    assert traces
    for path in traces:
        log = np.load(path)
        m, n, k, nenv, _ = (int(v) for v in log["geom"])
        wrap = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV))
        wrap.set_opponent(sp.OPP[path.split("_")[-2]]())
        replay_selfplay_trace(wrap, log, lambda w, sides: w.force_sides(torch.from_numpy(sides.astype(np.int64))))
    assert lib.jit_api_ready(4, 6, 3, lib.JIT_API_SP_PRE) and lib.jit_api_ready(4, 6, 3, lib.JIT_API_SP_POST)
    edges = np.load(f"{golden_dir}/edges.npz")
    played = 0
    for name, sc in sorted(SCENARIOS.items()):
        if (sc["m"], sc["n"], sc["k"]) in ((3, 3, 3), (9, 9, 5), (13, 13, 5), (15, 15, 5), (19, 19, 5)):
            continue
        got = play_scenario(hip.Env(sc["m"], sc["n"], sc["k"], 1, device=DEV), sc)
        assert np.array_equal(got, edges[name]), name
        played += 1
    assert played >= 3


@pytest.mark.parametrize("mode", ["generic", "specialised"])
def test_reference_fixtures_of_more_boards(hip, mode, golden_dir):
    """Round 4's fixtures (``tests/golden/make_golden.py --more-boards``: the imported reference's env op-logs on 5x5x4,
    6x7x4, 10x10x5, 12x12x5, 3x33x3 and wrapper traces on 6x7x4, 12x12x5, 5x5x4, 10x10x5 -- boards whose HIP kernels are
    compiled at run time) replayed on the generic kernels (MNK_JIT_API=0) and on the boards' own (MNK_JIT_API=1), with
    f32 and narrow observations: what the reference recorded, bit for bit, either way."""
    lib = hip.lib
    from replay import golden_files, replay_env_log, replay_selfplay_trace

    saved = os.environ.get("MNK_JIT_API")
    os.environ["MNK_JIT_API"] = "1" if mode == "specialised" else "0"
    lib.reload_config()
    try:
        envs, traces = golden_files(golden_dir, "boards_env_"), golden_files(golden_dir, "boards_selfplay_")
        assert len(envs) == 5 and len(traces) == 4
        for path in envs:
            log = np.load(path)
            m, n, k, nenv, _ = (int(v) for v in log["geom"])
            replay_env_log(hip.Env(m, n, k, nenv, device=DEV), log)
            replay_env_log(hip.Env(m, n, k, nenv, device=DEV, obs_dtype=torch.bfloat16), log)
            assert mode == "generic" or lib.jit_api_ready(m, n, k, lib.JIT_API_STEP), (m, n, k)
        for j, path in enumerate(traces):
            log = np.load(path)
            m, n, k, nenv, _ = (int(v) for v in log["geom"])
            wrap = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV, obs_dtype=[torch.float32, torch.uint8][j % 2]))
            wrap.set_opponent(sp.OPP[path.split("_")[-2]]())
            replay_selfplay_trace(wrap, log, lambda w, sides: w.force_sides(torch.from_numpy(sides.astype(np.int64))))
            assert mode == "generic" or lib.jit_api_ready(m, n, k, lib.JIT_API_SP_POST), (m, n, k)
    finally:
        if saved is None:
            os.environ.pop("MNK_JIT_API", None)
        else:
            os.environ["MNK_JIT_API"] = saved
        lib.reload_config()


@pytest.mark.parametrize("mode", ["generic", "specialised"])
def test_reference_learn_rollouts_on_a_board_without_a_built_in_variant(hip, mode, golden_dir):
    """Two consecutive ``PPOAgent.learn`` rollouts of the reference on 6x7x4 (``boards_ppo_learn_6x7x4_n64_t16.npz``:
    its own ``cnn_b_s``, sampling, ``RolloutBuffer``, GAE and one PPO update in between, recorded through a proxy),
    replayed on the HIP wrapper with the drop-in buffer attached as the sink -- on the generic kernels and on the
    board's own run-time compiled ones: every buffer field and the episode statistics equal the reference's."""
    lib = hip.lib
    from alg.rollout_buffer import RolloutBuffer
    from oracle.policies import MaskHashPolicy
    from replay import golden_files, replay_ppo_learn

    saved = os.environ.get("MNK_JIT_API")
    os.environ["MNK_JIT_API"] = "1" if mode == "specialised" else "0"
    lib.reload_config()
    try:
        (path,) = golden_files(golden_dir, "boards_ppo_learn_")
        log = np.load(path)
        m, n, k, nenv, n_steps = (int(v) for v in log["geom"])
        wrap = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV))
        wrap.set_opponent(MaskHashPolicy(1))
        wrap.track_episodes()
        buf = RolloutBuffer(n_steps, nenv, (2, m, n), m * n, device=DEV)
        wrap.attach_sink(buf)
        calls = []

        def make_buffer(*_):
            if calls:
                buf.reset()  # the end of the previous learn(): ppo.py:146
            calls.append(1)
            return buf

        replay_ppo_learn(wrap, make_buffer, log,
                         lambda w, sides: w.force_sides(torch.from_numpy(sides.astype(np.int64))),
                         episode_stats=wrap.pop_episode_stats)
        assert len(calls) == 2
        assert mode == "generic" or (lib.jit_api_ready(m, n, k, lib.JIT_API_SP_PRE) and lib.jit_api_ready(m, n, k, lib.JIT_API_SP_POST))
    finally:
        if saved is None:
            os.environ.pop("MNK_JIT_API", None)
        else:
            os.environ["MNK_JIT_API"] = saved
        lib.reload_config()


def _play(wrap, ora, steps, rng, where):
    """`steps` agent-steps of random legal moves on the HIP wrapper and on the oracle, compared after every step"""
    o1, _ = wrap.reset()
    o2, _ = ora.reset()
    for t in range(steps):
        mask = o2["action_mask"].numpy()
        acts = torch.from_numpy(np.array([rng.choice(np.nonzero(row)[0]) for row in mask]))
        o1, r1, t1, _, _ = wrap.step(acts.to(DEV))
        o2, r2, t2, _, _ = ora.step(acts)
        assert torch.equal(o1["observation"].cpu(), o2["observation"]), f"{where} step {t}"
        assert torch.equal(o1["action_mask"].cpu(), o2["action_mask"]), f"{where} step {t}"
        assert torch.equal(r1.cpu(), r2) and torch.equal(t1.cpu(), t2), f"{where} step {t}"


def test_a_kernel_switches_to_its_own_variant_once_it_is_hot(hip):
    """Default policy (no MNK_JIT_API): the generic kernels run until a kernel has been launched 1 024 times on the board
    (or has covered 2^26 items), then the board's own variant takes over -- in the middle of a game, with the same
    results (the oracle checks every step on both sides of the switch)."""
    lib = hip.lib
    for key in ("MNK_JIT_API", "MNK_JIT"):
        assert os.environ.get(key) is None, "this test needs the default policy"
    lib.reload_config()
    m, n, k, nenv = 8, 10, 4, 24  # a board no other test uses: its counters start at zero in this process
    wrap = hip.Wrapper(hip.Env(m, n, k, nenv, device=DEV), seed=3)
    sides = torch.zeros(nenv, dtype=torch.long)
    wrap.force_sides(sides)
    wrap.set_opponent(LowestLegalPolicy())
    ora = fz._ForcedSides(OracleVectorEnv(m, n, k, nenv))
    ora.sides = sides
    ora.set_opponent(LowestLegalPolicy())
    rng = np.random.default_rng(5)
    _play(wrap, ora, 1000, rng, "before the switch")  # reset = launch 1 of pre / post, then 1 000 steps
    assert not lib.jit_api_ready(m, n, k, lib.JIT_API_SP_PRE) and not lib.jit_api_ready(m, n, k, lib.JIT_API_SP_POST)
    _play(wrap, ora, 60, rng, "across the switch")    # (its reset + 22 steps reach launch 1 024)
    assert lib.jit_api_ready(m, n, k, lib.JIT_API_SP_PRE) and lib.jit_api_ready(m, n, k, lib.JIT_API_SP_POST)
    # 2^26 items: 128 launches over 2^19 envs
    big = hip.Env(6, 5, 4, 1 << 19, device=DEV)
    acts = torch.zeros(1 << 19, dtype=torch.long, device=DEV)
    rew = torch.empty(1 << 19, dtype=torch.float32, device=DEV)
    done = torch.empty(1 << 19, dtype=torch.bool, device=DEV)
    for t in range(127):
        big.step_into(acts + t % 30, rew, done)
    assert not lib.jit_api_ready(6, 5, 4, lib.JIT_API_STEP)
    for t in range(2):
        big.step_into(acts + t, rew, done)
    assert lib.jit_api_ready(6, 5, 4, lib.JIT_API_STEP)
    big.check_errors()


def test_nothing_is_compiled_under_a_capture_and_prepare_comes_before_it(hip, jit_api):
    """A launch whose stream is being captured into a hipGraph never compiles or loads a code object (either would
    invalidate the capture): it takes the generic kernel, and the graph replays to the same results.  ``jit_prepare``
    before the capture puts the board's own kernels into the graph."""
    lib = jit_api
    m, n, k, nenv = 7, 8, 4, 200
    c = m * n

    def rollout(prepare):
        env = hip.Env(m, n, k, nenv, device=DEV)
        wrap = hip.Wrapper(env, seed=11)
        from selfplay.policy import RandomPolicy

        wrap.set_opponent(RandomPolicy(c, seed=2))
        if prepare:
            assert env.specialise_kernels([lib.JIT_API_SP_STEP, lib.JIT_API_SAMPLE_LEGAL]) == 2
        out = {"observation": torch.empty((nenv, 2, m, n), dtype=torch.float32, device=DEV),
               "action_mask": torch.empty((nenv, c), dtype=torch.bool, device=DEV),
               "rewards": torch.empty(nenv, dtype=torch.float32, device=DEV),
               "terminated": torch.empty(nenv, dtype=torch.bool, device=DEV)}
        acts = torch.empty(nenv, dtype=torch.long, device=DEV)
        step_dev = torch.zeros(1, dtype=torch.int64, device=DEV)
        side = torch.cuda.Stream(DEV)
        side.wait_stream(torch.cuda.current_stream(DEV))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            wrap.reset(out=out)  # eager: not captured (step 0 of the wrapper's Philox stream)
        torch.cuda.current_stream(DEV).wait_stream(side)
        wrap.step_dev = step_dev  # from here on the step counter advances in device memory
        with torch.cuda.graph(graph):
            env.sample_legal_into(acts, seed=9, step=0, step_dev=step_dev)
            wrap._advance(acts, None, out=out)
            step_dev.add_(1)
        trace = []
        for _ in range(40):
            graph.replay()
            trace.append((out["observation"].clone(), out["action_mask"].clone(), out["rewards"].clone(),
                          out["terminated"].clone(), acts.clone()))
        env.check_errors()
        return trace

    ready = lambda kind: lib.jit_api_ready(m, n, k, kind)
    os.environ.pop("MNK_JIT_API")
    lib.reload_config()
    generic = rollout(prepare=False)  # the default policy, 41 launches: nothing is hot, everything generic
    assert not ready(lib.JIT_API_SP_STEP) and not ready(lib.JIT_API_SAMPLE_LEGAL)
    os.environ["MNK_JIT_API"] = "1"
    lib.reload_config()
    mixed = rollout(prepare=False)
    # the eager reset compiled the step kernel at its first launch; the draw's first launch was under the capture
    assert ready(lib.JIT_API_SP_STEP) and not ready(lib.JIT_API_SAMPLE_LEGAL), "a capture must not compile"
    special = rollout(prepare=True)
    assert ready(lib.JIT_API_SP_STEP) and ready(lib.JIT_API_SAMPLE_LEGAL)
    for t, (a, b, c3) in enumerate(zip(generic, mixed, special)):
        assert all(torch.equal(x, y) for x, y in zip(a, b)), f"replay {t} (generic / mixed)"
        assert all(torch.equal(x, y) for x, y in zip(a, c3)), f"replay {t} (generic / specialised)"
    assert any(bool(a[3].any()) for a in generic), "no game ended: the comparison saw no reset"
    assert lib.jit_prepare(9, 9, 5, [lib.JIT_API_SP_STEP]) == 0  # a built-in board: nothing to prepare


@pytest.mark.parametrize("agent,opponent,packed", [("net", "nn", False), ("net", "random", True), ("random", "random", False)])
def test_a_captured_rollout_runs_the_boards_own_kernels(hip, agent, opponent, packed):
    """``GraphedRollout`` on a board without a built-in variant, under the DEFAULT policy (nothing forced): its warm-up
    rollout launches every kernel a handful of times -- far from the 1 024 that make it hot -- and a capture cannot compile, so the constructor
    prepares the kernels the warm-up touched (``jit_prepare``) before it captures.  The graph then replays the
    specialised kernels, and fills the buffer like the eager loop, rollout after rollout."""
    lib = hip.lib
    for key in ("MNK_JIT_API", "MNK_JIT"):
        assert os.environ.get(key) is None, "this test needs the default policy"
    lib.reload_config()
    from alg.packed_rollout_buffer import PackedRolloutBuffer
    from alg.rollout_buffer import RolloutBuffer
    from selfplay import policy

    class NS:
        pass

    ns = NS()
    ns.lib, ns.Env, ns.Wrapper, ns.Buffer, ns.PackedBuffer, ns.policy = lib, hip.Env, hip.Wrapper, RolloutBuffer, PackedRolloutBuffer, policy
    board = {"nn": (6, 8, 4), "random": (7, 6, 4)}[opponent] if not packed else (8, 7, 5)  # boards no other test touches
    sk.graphed_rollout_against_the_eager_loop(ns, agent, opponent, packed, board)
    m, n, k = board
    logits = torch.float32 if agent == "net" else None
    if opponent == "random":
        assert lib.jit_api_ready(m, n, k, lib.jit_api_draw_kind(2, logits))   # the whole step, the agent's draw folded in
    else:
        assert lib.jit_api_ready(m, n, k, lib.jit_api_draw_kind(0, logits))   # pre with the agent's draw
        assert lib.jit_api_ready(m, n, k, lib.jit_api_draw_kind(1, torch.float32))  # post with the opponent's


@pytest.mark.parametrize("opponent", ["random", "nn"])
def test_a_captured_agent_step_runs_the_boards_own_kernels(hip, opponent):
    """the same for ``GraphedAgentStep`` (one agent-step per graph): prepared between its four warm-up steps and the
    capture, equal to the eager sequence (its twin wrapper comes later and finds the kernels prepared; the captured
    rollout test above compares specialised replays with a generic eager loop)"""
    lib = hip.lib
    for key in ("MNK_JIT_API", "MNK_JIT"):
        assert os.environ.get(key) is None, "this test needs the default policy"
    lib.reload_config()
    from selfplay import policy

    class NS:
        pass

    ns = NS()
    ns.lib, ns.Env, ns.Wrapper, ns.policy = lib, hip.Env, hip.Wrapper, policy
    board = (5, 8, 4) if opponent == "random" else (8, 5, 3)   # boards no other test touches
    sp.graphed_agent_step_against_eager(ns, opponent, board)
    m, n, k = board
    if opponent == "random":
        assert lib.jit_api_ready(m, n, k, lib.jit_api_draw_kind(2, torch.float32))
        assert lib.jit_api_ready(m, n, k, lib.JIT_API_SP_STEP)   # (the reset before the capture launched the plain form)
    else:
        assert lib.jit_api_ready(m, n, k, lib.jit_api_draw_kind(0, torch.float32))
        assert lib.jit_api_ready(m, n, k, lib.jit_api_draw_kind(1, torch.float32))


def test_records_and_minibatches_through_the_specialised_write_outs(hip, jit_api):
    """``mnk_unpack_records`` (records -> RolloutBuffer layout) and ``mnk_gather_obs`` (minibatch from packed planes) of
    a board without a built-in variant, specialised: == the generic kernels' output (MNK_JIT_API=0), which the records
    and the observations they are compared with elsewhere pin to the oracle; bf16 through the packed write-out too"""
    lib = jit_api
    from alg.packed_rollout_buffer import PackedRolloutBuffer

    m, n, k, nenv, t = 12, 12, 5, 333, 12

    def run():
        env = hip.Env(m, n, k, nenv, device=DEV)
        roll = hip.rollout.RandomRollout(env, seed=31)
        roll.run(30, record=False)
        rec = roll.run(t)
        fields = hip.rollout.unpack_records(rec, env)
        narrow = hip.rollout.unpack_records(rec, env, obs_dtype=torch.bfloat16)["observations"]
        # the same observations as packed planes (what a PackedRolloutBuffer stores), through the env's own packer
        buf = PackedRolloutBuffer(t, nenv, m, n, device=DEV)
        obs = fields["observations"]  # [t, N, 2, m, n] f32
        scratch = hip.Env(m, n, k, nenv, device=DEV)
        for j in range(t):
            scratch.boards = obs[j]
            buf.planes[j].copy_(scratch._planes)
        g = torch.Generator(device="cpu").manual_seed(1)
        idx = torch.randperm(t * nenv, generator=g)[:500].to(DEV)
        o, mk = buf.gather(idx)
        flat_obs = obs.reshape(t * nenv, 2, m, n)
        assert torch.equal(o, flat_obs[idx])  # (fix_empty only touches full boards: none after 42 plies of 144 cells)
        assert torch.equal(mk, fields["action_masks"].reshape(t * nenv, -1)[idx])
        return [fields[key].clone() for key in sorted(fields)] + [narrow.clone(), o.clone(), mk.clone()]

    os.environ["MNK_JIT_API"] = "0"
    lib.reload_config()
    want = run()
    os.environ["MNK_JIT_API"] = "1"
    lib.reload_config()
    got = run()
    assert lib.jit_api_ready(m, n, k, lib.JIT_API_GATHER_OBS) and lib.jit_api_ready(m, n, k, lib.JIT_API_UNPACK_RECORDS)
    for a, b in zip(want, got):
        assert torch.equal(a, b)


def test_a_later_process_loads_its_kernels_from_the_cache_on_disk(hip, tmp_path):
    """``$MNK_JIT_CACHE``: the second process to play on a board compiles nothing -- its kernels (rollout and API) come
    from the code objects the first one left on disk -- and plays the same games."""
    import json
    import subprocess
    import sys

    script = r'''
import hashlib, json, os, sys
sys.path[:0] = [os.environ["MNK_ROOT"], os.path.join(os.environ["MNK_ROOT"], "rl-selfplay-mnk_amd")]
import torch
import mnk_hip
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from selfplay.policy import RandomPolicy
from selfplay.random_rollout import RandomRollout
from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper
env = TorchVectorMnkEnv(8, 11, 4, 1 << 13, device="cuda:0")
rec = RandomRollout(env, seed=3).run(128)          # 2^20 env-steps: the rollout kernel is specialised
wrap = TorchSelfPlayWrapper(env, seed=5)
wrap.set_opponent(RandomPolicy(88, seed=7))
obs, _ = wrap.reset()
h = hashlib.sha256(rec.planes.cpu().numpy().tobytes() + rec.meta.cpu().numpy().tobytes())
for t in range(6):
    obs, rew, term, _, _ = wrap.step(torch.full((1 << 13,), t, dtype=torch.long, device="cuda:0"))
    h.update(obs["observation"].cpu().numpy().tobytes() + rew.cpu().numpy().tobytes() + term.cpu().numpy().tobytes())
env.check_errors()
print(json.dumps({"digest": h.hexdigest(), "stats": mnk_hip.jit_stats(),
                  "ready": mnk_hip.jit_api_ready(8, 11, 4, mnk_hip.JIT_API_SP_STEP)}))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MNK_JIT_CACHE=str(tmp_path / "cache"), MNK_JIT_API="1", MNK_ROOT=root)
    runs = []
    for _ in range(2):
        res = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, env=env)
        assert res.returncode == 0, res.stderr[-2000:]
        runs.append(json.loads(res.stdout.strip().splitlines()[-1]))
    first, second = runs
    assert first["ready"] and second["ready"]
    assert first["stats"]["compiled"] >= 2 and first["stats"]["cache_stores"] == first["stats"]["compiled"]
    assert second["stats"]["compiled"] == 0 and second["stats"]["cache_hits"] == first["stats"]["compiled"], second
    assert first["digest"] == second["digest"]
