"""GPU parity: the HIP env (through the C ABI) against the golden vectors of the
reference and against the oracle.  Bit-exact everywhere (integer / byte work)."""
import os

import numpy as np
import pytest
import torch

from oracle import philox
from oracle.env_torch import OracleVectorEnv
from oracle.packing import pack_boards, record_rows
from oracle.rollout import encode_action_log, random_rollout
from oracle.rollout import replay_actions as oracle_replay
from conftest import random_play_stats
from replay import golden_files, play_scenario, replay_env_log
from scenarios import SCENARIOS

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    import __graft_entry__ as entry

    entry.build_hip()
    entry._ensure_path()
    import mnk_hip
    from env.torch_vector_mnk_env import TorchVectorMnkEnv
    from selfplay import random_rollout
    from selfplay.random_rollout import RandomRollout

    mnk_hip.load()
    assert torch.cuda.is_available()

    class NS:
        pass

    ns = NS()
    ns.lib, ns.Env, ns.Rollout, ns.rollout = mnk_hip, TorchVectorMnkEnv, RandomRollout, random_rollout
    return ns


@pytest.mark.parametrize("idx", range(11))
def test_env_oplog_matches_reference(hip, golden_dir, idx):
    """G1 + G2: step / step_subset / reset(idx) op-logs recorded from the reference."""
    log = np.load(golden_files(golden_dir, "env_")[idx])
    m, n, k, nenv, _ = (int(v) for v in log["geom"])
    replay_env_log(hip.Env(m, n, k, nenv, device=DEV), log)


@pytest.mark.parametrize("obs_dtype", [torch.bfloat16, torch.uint8])
@pytest.mark.parametrize("idx", [0, 3, 5, 8, 10])
def test_env_oplog_matches_reference_with_narrow_observations(hip, golden_dir, idx, obs_dtype):
    """The same reference op-logs on an env that hands out bf16 / u8 observations (ABI v4, opt-in): the cells the
    reference recorded, in the narrow dtype."""
    log = np.load(golden_files(golden_dir, "env_")[idx])
    m, n, k, nenv, _ = (int(v) for v in log["geom"])
    replay_env_log(hip.Env(m, n, k, nenv, device=DEV, obs_dtype=obs_dtype), log)


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_edge_scenarios_match_reference(hip, golden_dir, name):
    """G4: poked positions, answers recorded from the reference."""
    sc = SCENARIOS[name]
    want = np.load(f"{golden_dir}/edges.npz")[name]
    got = play_scenario(hip.Env(sc["m"], sc["n"], sc["k"], 1, device=DEV), sc)
    assert np.array_equal(got, want)


def test_reference_test_env_mechanics_win(hip):
    """src/tests/test_mnk_integration.py:50-65, statement for statement."""
    env = hip.Env(m=3, n=3, k=3, num_envs=1, device=DEV)
    env.reset()
    env.boards[0, 0, 0, 0] = 1
    env.boards[0, 0, 0, 1] = 1
    actions = torch.tensor([2], device=env.device)
    _, rewards, dones = env.step(actions)
    assert dones[0].item() is True
    assert rewards[0].item() == 1.0


def test_reference_test_env_illegal_move_strict(hip):
    """src/tests/test_mnk_integration.py:68-81 -- fails on the reference as shipped (its validator is
    dead code); here the check is opt-in (strict=True) and the default stays bit-compatible."""
    env = hip.Env(m=3, n=3, k=3, num_envs=1, device=DEV, strict=True)
    env.reset()
    env.boards[0, 0, 0, 0] = 1
    with pytest.raises(ValueError, match="Illegal Move"):
        env.step(torch.tensor([0], device=env.device))
    # the refused move left the env untouched
    assert env.boards.sum().item() == 1.0 and int(env.move_counts[0]) == 0


def test_out_of_range_action_is_reported(hip):
    env = hip.Env(3, 3, 3, 4, device=DEV)
    env.reset()
    env.step(torch.tensor([0, 9, 1, 2], device=DEV))
    with pytest.raises(IndexError):
        env.check_errors()
    # env 1 untouched, the others moved
    assert env.move_counts.tolist() == [1, 0, 1, 1]


def test_views_behave_like_dense_tensors(hip):
    env = hip.Env(4, 6, 3, 8, device=DEV)
    env.reset()
    env.boards[2, 1, 3, 5] = 1.0
    env.boards[3, 0] = torch.ones(4, 6)
    assert env.boards.shape == (8, 2, 4, 6) and env.boards.dtype == torch.float32
    assert env.boards[2].sum().item() == 1.0 and env.boards[3, 0].sum().item() == 24.0
    assert float(torch.sum(env.boards)) == 25.0
    assert env.boards[0].cpu().numpy().shape == (2, 4, 6)
    env.current_player[torch.tensor([1, 5], device=DEV)] = 1
    assert env.current_player.tolist() == [0, 1, 0, 0, 0, 1, 0, 0]
    assert (env.current_player == 1).sum().item() == 2
    idx = torch.tensor([5, 6], device=DEV)
    env.current_player[idx] ^= 1
    assert env.current_player.tolist() == [0, 1, 0, 0, 0, 0, 1, 0]
    env.move_counts[4] = 7
    assert env.move_counts.tolist() == [0, 0, 0, 0, 7, 0, 0, 0]
    assert env.current_player.tolist() == [0, 1, 0, 0, 0, 0, 1, 0]  # untouched by the count write
    mask = env.observe()["action_mask"]
    assert mask[2].sum().item() == 23 and mask[3].sum().item() == 0
    env.boards.zero_()
    assert env.boards.sum().item() == 0.0
    env.reset(torch.tensor([], dtype=torch.long, device=DEV))  # reset(empty) is a no-op


@pytest.mark.parametrize("m,n,k,nenv,flip", [(3, 3, 3, 5, False), (9, 9, 5, 130, True), (19, 19, 5, 67, True),
                                             (4, 6, 3, 64, True), (13, 13, 5, 1, False)])
def test_observe_flip_and_empty_mask_fix(hip, m, n, k, nenv, flip):
    """mnk_observe against numpy: absolute and canonical view (wrapper:99-112), ragged sizes."""
    rng = np.random.default_rng(m * 100 + n)
    dense = (rng.random((nenv, 2, m, n)) < 0.45).astype(np.float32)
    dense[0] = 1.0  # a full board: no legal cell
    env = hip.Env(m, n, k, nenv, device=DEV)
    env.boards = torch.from_numpy(dense)
    obs = env.observe()
    assert np.array_equal(obs["observation"].cpu().numpy(), dense)
    legal = ~(dense != 0).any(axis=1).reshape(nenv, m * n)
    assert np.array_equal(obs["action_mask"].cpu().numpy(), legal)
    if flip:
        side = torch.from_numpy(rng.integers(0, 2, nenv)).to(DEV)
        o = torch.empty((nenv, 2, m, n), dtype=torch.float32, device=DEV)
        msk = torch.empty((nenv, m * n), dtype=torch.bool, device=DEV)
        env.observe_into(o, msk, flip_side=side, fix_empty_mask=True)
        want = dense.copy()
        s = side.cpu().numpy() == 1
        want[s] = want[s][:, ::-1]
        assert np.array_equal(o.cpu().numpy(), want)
        legal2 = legal.copy()
        legal2[legal2.sum(axis=1) == 0, 0] = True
        assert np.array_equal(msk.cpu().numpy(), legal2)


@pytest.mark.parametrize("m,n,k,nenv", [(3, 3, 3, 64), (9, 9, 5, 1000), (19, 19, 5, 130), (13, 13, 5, 65),
                                        (4, 6, 3, 33), (22, 22, 5, 16)])
def test_sample_legal_matches_oracle(hip, m, n, k, nenv):
    """mnk_sample_legal == oracle Philox + pick_legal, bit for bit (RandomPolicy stand-in)."""
    rng = np.random.default_rng(7)
    dense = (rng.random((nenv, 2, m, n)) < 0.35).astype(np.float32)
    dense[1] = 1.0  # no legal cell -> uniform over all cells
    env = hip.Env(m, n, k, nenv, device=DEV)
    env.boards = torch.from_numpy(dense)
    legal = ~(dense != 0).any(axis=1).reshape(nenv, m * n)
    acts = torch.empty(nenv, dtype=torch.int64, device=DEV)
    for step, stream, id0 in [(0, 0, 0), (5, 1, 1000), (2 ** 33 + 3, 2, 2 ** 32 + 5)]:
        env.sample_legal_into(acts, seed=99, step=step, env_id0=id0, stream_id=stream)
        x = philox.rand_u32(99, np.arange(id0, id0 + nenv, dtype=np.uint64), step, stream)
        assert np.array_equal(acts.cpu().numpy(), philox.pick_legal(legal, x))


def _force_form(param):
    saved = {key: os.environ.get(key) for key in ("MNK_ROLLOUT_PAIR", "MNK_ROLLOUT_FORM")}
    os.environ.pop("MNK_ROLLOUT_FORM", None)
    os.environ["MNK_ROLLOUT_PAIR"] = "1" if param.startswith("two lanes") else "0"
    if "waves" in param:
        os.environ["MNK_ROLLOUT_FORM"] = "ws2" if param.startswith("two") else "ws4"
    elif param.startswith("two lanes"):  # split by scan directions, or by board words (19x19 / 15x15 only)
        os.environ["MNK_ROLLOUT_FORM"] = "pairw" if "words" in param else "pair"
    _reload_knobs()
    return saved


def _reload_knobs():
    """the library reads its environment knobs once; tell it the environment has changed"""
    import __graft_entry__ as entry

    entry._ensure_path()
    import mnk_hip

    mnk_hip.reload_config()


def _restore_form(saved):
    for key, val in saved.items():
        if val is None:
            os.environ.pop(key, None)
        else:
            os.environ[key] = val
    _reload_knobs()


@pytest.fixture(params=["one lane per env", "two lanes per env", "two lanes per env, words split"])
def lane_or_pair(request):
    """the two forms that can write the action log"""
    saved = _force_form(request.param)
    yield request.param
    _restore_form(saved)


@pytest.fixture(params=["one lane per env", "two lanes per env", "two lanes per env, words split",
                        "two waves per env group", "four waves per env group"])
def lanes_per_env(request):
    """The launcher picks the rollout kernel form by board and batch size; MNK_ROLLOUT_PAIR / MNK_ROLLOUT_FORM (read
    on every call) force one, so small test batches reach every form.  Boards without a compile-time specialisation
    have the one-lane form only; the waves-per-group forms exist for 9x9x5 and 19x19x5 (others fall through to the
    launcher's own choice)."""
    saved = _force_form(request.param)
    yield request.param
    _restore_form(saved)


@pytest.mark.parametrize("m,n,k", [(3, 3, 3), (9, 9, 5), (19, 19, 5), (7, 9, 7), (22, 22, 5)])
def test_sample_legal_reaches_every_cell_and_every_rank(hip, m, n, k):
    """The r-th-set-bit select at its edges: one env per cell with only that cell free (the draw must be that
    cell whatever the random word), and boards with a known set of free cells where every rank is hit --
    first and last cell, word boundaries of the bit string -- against the oracle's pick_legal."""
    c = m * n
    dense = np.ones((c, 2, m, n), dtype=np.float32)
    dense[:, 1] = 0.0
    for cell in range(c):
        dense[cell, 0, cell // n, cell % n] = 0.0
    env = hip.Env(m, n, k, c, device=DEV)
    env.boards = torch.from_numpy(dense)
    acts = torch.empty(c, dtype=torch.int64, device=DEV)
    for step in (0, 1, 2, 3, 12345):
        env.sample_legal_into(acts, seed=step, step=step, env_id0=0, stream_id=0)
        assert acts.cpu().tolist() == list(range(c))
    # free cells at the corners and around the 32-bit word boundaries of the guard-column bit string
    bits = sorted({0, 1, 30, 31, 32, 33, 62, 63, 64, 65, 94, 95, 96, 97, m * (n + 1) - 2})
    cells = sorted({b - b // (n + 1) for b in bits if b % (n + 1) != n and b < m * (n + 1) - 1} | {0, c - 1})
    nenv = 4096
    dense = np.ones((nenv, 2, m, n), dtype=np.float32)
    dense[:, 1] = 0.0
    for cell in cells:
        dense[:, 0, cell // n, cell % n] = 0.0
    env = hip.Env(m, n, k, nenv, device=DEV)
    env.boards = torch.from_numpy(dense)
    acts = torch.empty(nenv, dtype=torch.int64, device=DEV)
    env.sample_legal_into(acts, seed=7, step=9, env_id0=100, stream_id=0)
    legal = ~(dense != 0).any(axis=1).reshape(nenv, c)
    x = philox.rand_u32(7, np.arange(100, 100 + nenv, dtype=np.uint64), 9, 0)
    got = acts.cpu().numpy()
    assert np.array_equal(got, philox.pick_legal(legal, x))
    assert sorted(set(got.tolist())) == cells  # 4096 draws over <= 16 cells: every rank occurs


@pytest.mark.parametrize("m,n,k,nenv,chunks", [(3, 3, 3, 64, (7, 9, 16)), (9, 9, 5, 333, (64, 31)),
                                               (3, 3, 3, 3, (30,)), (9, 9, 5, 1, (5, 6, 200)), (5, 6, 4, 129, (61,)),
                                               (4, 6, 3, 100, (40,)), (13, 13, 5, 65, (120,)),
                                               (19, 19, 5, 64, (200,)), (7, 9, 7, 70, (90,)),
                                               (15, 15, 5, 33, (37, 130)), (19, 19, 5, 31, (3, 9, 190))])
def test_rollout_matches_oracle(hip, m, n, k, nenv, chunks, lanes_per_env):
    """mnk_rollout_random == oracle loop (sample -> step -> reset done), records, stats and final
    state bit for bit; several launches continue the same Philox step counter.  Both kernel forms: one lane per
    env (what large batches, the headline size included, run) and two lanes per env (small batches)."""
    env = hip.Env(m, n, k, nenv, device=DEV)
    roll = hip.Rollout(env, seed=5, env_id0=12345)
    ora = OracleVectorEnv(m, n, k, nenv)
    total = np.zeros(5, dtype=np.int64)
    step0 = 0
    for t in chunks:
        rec = roll.run(t)
        planes, meta, stats = random_rollout(ora, seed=5, step0=step0, steps=t, env_id0=12345)
        total += stats
        step0 += t
        assert np.array_equal(rec.planes.cpu().numpy().view(np.uint64), planes)
        assert np.array_equal(rec.meta.cpu().numpy().view(np.uint32), meta)
        assert np.array_equal(roll.stats.cpu().numpy(), total)
        assert np.array_equal(pack_boards(env.boards.cpu().numpy(), m, n), pack_boards(ora.boards.numpy(), m, n))
        assert np.array_equal(env.current_player.cpu().numpy(), ora.current_player.numpy())
        assert np.array_equal(env.move_counts.cpu().numpy(), ora.move_counts.numpy())


@pytest.mark.parametrize("m,n,k", [(9, 9, 5), (3, 3, 3), (13, 13, 5), (4, 6, 3), (15, 15, 5)])
def test_rollout_on_poked_states_takes_the_general_loop(hip, m, n, k):
    """The one-lane rollout kernel plays a wave whose games are all CONSISTENT (stones on the board == plies counted
    < C: every state the env itself produces) on a loop without the full-board branch and without a ply counter
    (mnk_rollout_lane.h, FAST); a wave that holds a poked state plays the general loop.  Here 256 envs = four waves:
    the first two untouched, the other two carrying hand-made states -- a full board with the counter at 0 (every ply
    lands on an occupied cell until the counter reaches C), a counter ahead of / behind the stones, two stones on one
    cell, a finished game left un-reset (counter == C), a position one ply short of a draw -- and everything equals
    the oracle, records, statistics and final state, over several launches."""
    nenv, c = 256, m * n
    env = hip.Env(m, n, k, nenv, device=DEV)
    ora = OracleVectorEnv(m, n, k, nenv)
    # mid-game positions everywhere first (both sides identical), then the pokes
    hip.Rollout(env, seed=9).run(max(4, c // 3) - max(4, c // 3) % 4, record=False)
    random_rollout(ora, seed=9, step0=0, steps=max(4, c // 3) - max(4, c // 3) % 4)
    assert np.array_equal(pack_boards(env.boards.cpu().numpy(), m, n), pack_boards(ora.boards.numpy(), m, n))
    rng = np.random.default_rng(5)

    def poke(i, boards, side, count):
        for target in (env, ora):
            if boards is not None:
                target.boards[i] = torch.from_numpy(boards)
            target.current_player[i] = side
            target.move_counts[i] = count

    full = np.zeros((2, m, n), dtype=np.float32)
    full[0].reshape(-1)[0::2] = 1.0
    full[1].reshape(-1)[1::2] = 1.0           # a full board (it may or may not hold a line: the scan decides)
    poke(130, full, 0, 0)                     # full, counter 0: occupied-cell plies until the counter reaches C
    poke(131, full, 1, c - 2)                 # full, two plies from the draw by the counter
    poke(140, None, 0, 3 * c)                 # counter far ahead of the stones: "draw" at the next ply without a win
    poke(141, None, 1, 0)                     # counter behind the stones
    both = np.zeros((2, m, n), dtype=np.float32)
    both[:, 0, 0] = 1.0                       # the same cell taken by both sides (an occupied-cell overwrite happened)
    both[0, 1, 1] = 1.0
    poke(200, both, 0, 3)
    sparse = (rng.random((2, m, n)) < 0.15).astype(np.float32)
    sparse[1] *= 1.0 - sparse[0]
    poke(201, sparse, 1, c)                   # a finished game that was never reset: counter == C already
    poke(255, sparse, 0, int(sparse.sum()))   # consistent by construction, in a wave that is not
    assert torch.equal(env.move_counts.cpu(), ora.move_counts)
    roll = hip.Rollout(env, seed=77, env_id0=1000)
    total = np.zeros(5, dtype=np.int64)
    step0 = 0
    for t in (8, c + 4, 20, 13):
        rec = roll.run(t)
        planes, meta, stats = random_rollout(ora, seed=77, step0=step0, steps=t, env_id0=1000)
        total += stats
        step0 += t
        assert np.array_equal(rec.planes.cpu().numpy().view(np.uint64), planes), t
        assert np.array_equal(rec.meta.cpu().numpy().view(np.uint32), meta), t
        assert np.array_equal(roll.stats.cpu().numpy(), total), t
        assert np.array_equal(pack_boards(env.boards.cpu().numpy(), m, n), pack_boards(ora.boards.numpy(), m, n)), t
        assert np.array_equal(env.current_player.cpu().numpy(), ora.current_player.numpy()), t
        assert np.array_equal(env.move_counts.cpu().numpy(), ora.move_counts.numpy()), t


def test_rollout_is_independent_of_sharding(hip):
    """Global env ids key the RNG: two shards of 96 envs reproduce one batch of 192."""
    m, n, k = 9, 9, 5
    whole = hip.Rollout(hip.Env(m, n, k, 192, device=DEV), seed=3).run(70)
    parts = [hip.Rollout(hip.Env(m, n, k, 96, device=DEV), seed=3, env_id0=96 * r).run(70) for r in (0, 1)]
    assert torch.equal(whole.planes, torch.cat([p.planes for p in parts], dim=2))
    assert torch.equal(whole.meta, torch.cat([p.meta for p in parts], dim=1))


@pytest.mark.parametrize("m,n,k,nenv,warm,plies", [
    (9, 9, 5, 65536, 152, 12),      # BASELINE configs 2 / 4: one lane per env, the FAST loop
    (19, 19, 5, 32768, 400, 4),     # BASELINE config 5's per-GPU batch: two lanes per env, board split by words
    (13, 13, 5, 65536, 200, 4),
])
def test_full_size_rollout_equals_the_oracle_ply_for_ply(hip, m, n, k, nenv, warm, plies):
    """Bit-exact parity AT the BASELINE.json batch sizes, not only properties: the device plays `warm` plies (a
    stationary mix of game phases, finished games restarting), the whole state goes to the CPU oracle, and then both
    play the same `plies` plies -- the oracle's raw loop (RandomPolicy with the Philox draw -> env.step ->
    env.reset(done)) against ONE launch of the fused kernel, and against the same plies through mnk_step_random:
    records, statistics and the final state of all 65 536 / 32 768 envs equal."""
    env = hip.Env(m, n, k, nenv, device=DEV)
    roll = hip.Rollout(env, seed=3, env_id0=1 << 20)
    roll.run(warm, record=False)
    ora = OracleVectorEnv(m, n, k, nenv)
    ora.boards.copy_(env.boards.cpu())
    ora.current_player.copy_(env.current_player.cpu())
    ora.move_counts.copy_(env.move_counts.cpu())
    twin = hip.Env(m, n, k, nenv, device=DEV)
    twin.load_state_dict(env.state_dict())
    before = roll.stats.cpu().numpy().copy()
    rec = roll.run(plies)
    planes, meta, stats = random_rollout(ora, seed=3, step0=warm, steps=plies, env_id0=1 << 20)
    assert int(stats[0]) > 0, "no game finished in the compared plies: lengthen them"
    assert np.array_equal(rec.planes.cpu().numpy().view(np.uint64), planes)
    assert np.array_equal(rec.meta.cpu().numpy().view(np.uint32), meta)
    assert np.array_equal(roll.stats.cpu().numpy() - before, stats)
    assert np.array_equal(pack_boards(env.boards.cpu().numpy(), m, n), pack_boards(ora.boards.numpy(), m, n))
    assert torch.equal(env.current_player.cpu(), ora.current_player) and torch.equal(env.move_counts.cpu(), ora.move_counts)
    # the same plies, one launch each
    rew = torch.empty(nenv, dtype=torch.float32, device=DEV)
    done = torch.empty(nenv, dtype=torch.bool, device=DEV)
    acts = torch.empty(nenv, dtype=torch.long, device=DEV)
    mask = torch.empty((nenv, m * n), dtype=torch.bool, device=DEV)
    for j in range(plies):
        twin.step_random_into(rew, done, mask, actions=acts, seed=3, step=warm + j, env_id0=1 << 20)
        assert np.array_equal(acts.cpu().numpy(), (meta[j] & 0xFFFF).astype(np.int64)), j
        assert np.array_equal(done.cpu().numpy(), ((meta[j] >> 24) & 1).astype(bool)), j
    assert torch.equal(twin._planes, env._planes) and torch.equal(twin._meta, env._meta)
    assert torch.equal(mask.cpu(), ora.observe()["action_mask"])


@pytest.mark.parametrize("m,n,k,nenv,steps,tail", [
    (9, 9, 5, 65536, 160, 640),       # BASELINE configs 2 / 4: one lane per env
    (19, 19, 5, 32768, 120, 2400),    # BASELINE config 5's per-GPU batch: two lanes per env
    (9, 9, 5, 1 << 20, 24, 776),      # 16 x the headline batch: 64-bit indexing, 16 waves per SIMD
    (12, 12, 5, 65536, 64, 1216),     # no built-in variant: the run-time specialised kernel
])
def test_full_size_rollout_properties(hip, m, n, k, nenv, steps, tail):
    """BASELINE.json sizes: properties that need no oracle run -- every recorded action was legal on the recorded
    board, the next board is the previous one plus that stone (or empty after a finished game), and the statistics of
    uniform random play sit on the reference's (tests/golden/random_play_stats.npz: first games of the imported
    reference's env + RandomPolicy, made by tests/golden/make_golden_stats.py)."""
    env = hip.Env(m, n, k, nenv, device=DEV)
    roll = hip.Rollout(env, seed=1)
    rec = roll.run(steps)
    rows = rec.planes  # [T,R,N] int64: mover's word | other side's word << 32
    side = rec.sides()
    done = rec.dones()
    lo, hi = rows & 0xFFFFFFFF, (rows >> 32) & 0xFFFFFFFF
    white_moves = (side == 1).unsqueeze(1)
    planes = torch.stack([torch.where(white_moves, hi, lo), torch.where(white_moves, lo, hi)], dim=1)  # [T,2,R,N] absolute
    del lo, hi
    act = rec.actions()
    bit = act + act // n
    word, sh = bit >> 5, bit & 31
    occ = planes[:, 0] | planes[:, 1]  # [T,R,N]
    occ_at = torch.gather(occ, 1, word.unsqueeze(1)).squeeze(1)
    assert not bool(((occ_at >> sh) & 1).any()), "an occupied cell was played"
    stone = torch.zeros_like(planes[:-1])
    one = (torch.ones_like(sh) << sh)[:-1]
    for p in (0, 1):
        for w in range(planes.shape[2]):
            stone[:, p, w] = torch.where((side[:-1] == p) & (word[:-1] == w), one, torch.zeros_like(one))
    expect = torch.where(done[:-1].unsqueeze(1).unsqueeze(1), torch.zeros_like(stone), planes[:-1] | stone)
    assert torch.equal(planes[1:], expect)
    del rows
    del planes, occ, stone, expect, one
    roll.run(tail, record=False)  # a longer window so games cut off at its end do not bias the mean
    episodes, black, white, draws, length = roll.stats.tolist()
    assert episodes == black + white + draws and episodes > 400000
    # A window of `steps + tail` plies after a common start drops the game in flight at its end, which is long on
    # average (inspection paradox): the games COUNTED are short-biased by about mean_length / window of the length's
    # relative variance -- sd^2 / window plies -- and the longest games (draws) are under-counted accordingly.  The
    # unbiased check, over 1000 x 256 plies, is test_rollout_soak_is_deterministic.
    ref = random_play_stats(f"{m}x{n}x{k}")
    window = steps + tail
    bias = ref["sd_plies"] ** 2 / window
    mean = length / episodes
    assert ref["mean_plies"] - 2.0 * bias - 5 * ref["se_mean_plies"] < mean < ref["mean_plies"] + 5 * ref["se_mean_plies"], \
        (mean, ref["mean_plies"], bias)
    rate = draws / episodes
    assert 0.6 * ref["draw_rate"] - 5 * ref["se_draw_rate"] <= rate <= ref["draw_rate"] + 5 * ref["se_draw_rate"] + 5 * (ref["draw_rate"] / episodes) ** 0.5
    assert black > white  # first-move advantage


@pytest.mark.parametrize("m,n,k,nenv,steps", [(3, 3, 3, 70, 40), (9, 9, 5, 200, 130), (19, 19, 5, 65, 90),
                                              (13, 13, 5, 5, 150), (7, 9, 7, 64, 81), (9, 9, 5, 64, 6),
                                              (15, 15, 5, 40, 120), (11, 11, 5, 33, 70)])
def test_action_log_replay_rebuilds_the_records(hip, m, n, k, nenv, steps, lane_or_pair):
    """The multi-GPU exchange formats: the action log as bytes, 16-bit fields or a 7-bit stream (boards of up to 128
    cells), in a self-contained message (chunk-start state + log) or alone (the receiver keeps the replay state).
    The log == the oracle's packing of the recorded actions; mnk_replay_actions on it == the records the rollout
    wrote (bit for bit) == the oracle's replay of the same actions; a second chunk checks that the state carried over."""
    from selfplay.random_rollout import (ACT_BITS7, ACT_U8, ACT_U8P1, ACT_U16, GatheredLogs, ReplayState, action_log_fits,
                                         action_log_format, action_log_words, replay_shard, unpack_action_log)

    c = m * n
    assert action_log_format(c) == (ACT_BITS7 if c <= 128 else ACT_U8 if c <= 256 else ACT_U8P1)
    assert action_log_format(c, compact=False) == (ACT_U8 if c <= 256 else ACT_U16)
    formats = [f for f in (ACT_U8, ACT_U16, ACT_BITS7, ACT_U8P1) if action_log_fits(f, c)]
    oracle_records = None
    for fmt in formats:
        for with_state in (True, False):
            env = hip.Env(m, n, k, nenv, device=DEV)
            roll = hip.Rollout(env, seed=21, env_id0=7)
            state = None if with_state else hip.rollout.gather_start_state(env)   # one rank: a copy of the state
            assert with_state or isinstance(state, ReplayState)
            got = []
            # with a log every chunk but the last is a multiple of four plies (one group of the log = plies 4q..4q+3)
            for chunk in (steps - steps % 4, steps):
                rec = roll.alloc(chunk, log_actions=fmt, with_state=with_state)
                assert rec.fmt == fmt and (rec.planes0 is not None) == with_state
                roll.run(chunk, out=rec)
                if fmt == ACT_U16:
                    assert rec.act.dtype == torch.int64 and rec.act.shape == ((chunk + 3) // 4, nenv)
                else:
                    assert rec.act.dtype == torch.int32 and rec.act.shape == (action_log_words(fmt, chunk), nenv)
                    assert hip.lib.load().mnk_action_log_words(fmt, chunk) == rec.act.shape[0]
                want_log = encode_action_log(rec.actions().cpu().numpy(), fmt)
                assert np.array_equal(rec.act.cpu().numpy().view(want_log.dtype), want_log), (fmt, chunk)
                assert torch.equal(unpack_action_log(rec.act, chunk, fmt), rec.actions())
                if with_state:  # the chunk-start state travels with the log (state layout; the first record is the same position)
                    assert np.array_equal(record_rows(rec.planes0.cpu().numpy().view(np.uint64), m, n, rec.meta0.cpu().numpy() & 1),
                                          rec.planes[0].cpu().numpy().view(np.uint64))
                logs = GatheredLogs.empty(1, env.words, nenv, chunk, c, DEV, fmt=fmt, with_state=with_state)
                assert logs.msg.shape == (1, rec.msg.numel())
                logs.msg.copy_(rec.msg.unsqueeze(0))  # what a one-rank all-gather delivers
                again = replay_shard(logs, 0, m, n, k, state=state)
                assert torch.equal(again.planes, rec.planes) and torch.equal(again.meta, rec.meta), (fmt, with_state, chunk)
                if state is not None:  # the receiver's replay state has moved on with the sender's env
                    assert torch.equal(state.planes[0], env._planes) and torch.equal(state.meta[0], env._meta)
                got.append((rec.planes.cpu().numpy().view(np.uint64), rec.meta.cpu().numpy().view(np.uint32)))
            if oracle_records is None:  # the same rollout whatever the log format: one oracle replay serves them all
                ora = OracleVectorEnv(m, n, k, nenv)
                oracle_records = [oracle_replay(ora, (meta & 0xFFFF).astype(np.int64)) for _, meta in got]
            for (planes, meta), (wp, wm) in zip(got, oracle_records):
                assert np.array_equal(planes, wp) and np.array_equal(meta, wm), (fmt, with_state)
    # 0.875 B per env-step against 1 B up to 128 cells; 1.125 B against 2 B above 256
    assert action_log_words(ACT_BITS7, 256) * 4 == 224 and action_log_words(ACT_U8, 256) * 4 == 256
    assert action_log_words(ACT_U8P1, 256) * 4 == 288 and action_log_words(ACT_U16, 256) * 4 == 512
    with pytest.raises(ValueError):
        hip.Rollout(hip.Env(13, 13, 5, 4, device=DEV)).alloc(8, log_actions=ACT_BITS7)  # 169 cells do not fit 7 bits


@pytest.mark.parametrize("m,n,k,nenv", [(9, 9, 5, 300), (19, 19, 5, 70)])
def test_keyframed_log_stream_rebuilds_any_chunk_on_demand(hip, m, n, k, nenv):
    """The default exchange of the sharded rollout: every K-th chunk's message carries the chunk-start state (a
    keyframe), the others the log alone.  ``KeyframedLogs`` on the receiving side rebuilds the records of ANY chunk
    held since the last keyframe -- state-only replays up to it, then a recording one -- bit-identical to what the
    sender recorded, and a new keyframe drops the history before it."""
    from selfplay.random_rollout import GatheredLogs, KeyframedLogs

    every, chunk = 3, 20
    env = hip.Env(m, n, k, nenv, device=DEV)
    roll = hip.Rollout(env, seed=31, env_id0=500)
    history = KeyframedLogs(m, n, k)
    with pytest.raises(ValueError):
        history.push(GatheredLogs(planes0=None, meta0=None, act=torch.zeros((1, 1, nenv), dtype=torch.int32, device=DEV), steps=4))
    sent = []
    for c in range(8):
        key = c % every == 0
        rec = roll.alloc(chunk, log_actions=True, with_state=key)
        roll.run(chunk, out=rec)
        logs = GatheredLogs.empty(1, env.words if key else 0, nenv, chunk, m * n, DEV, fmt=rec.fmt, with_state=key)
        logs.msg.copy_(rec.msg.unsqueeze(0))   # what a one-rank all-gather delivers
        history.push(logs)
        logs.msg.zero_()                        # the gather buffer is reused: the history holds its own copy
        sent = [rec] if key else sent + [rec]
        assert history.chunks() == len(sent) == c % every + 1
        for j, want in enumerate(sent):
            got = history.rebuild(0, j)
            assert torch.equal(got.planes, want.planes) and torch.equal(got.meta, want.meta), (c, j)
    with pytest.raises(IndexError):
        history.rebuild(0, history.chunks())


def test_action_log_needs_aligned_chunks(hip):
    env = hip.Env(3, 3, 3, 8, device=DEV)
    roll = hip.Rollout(env, seed=1)
    roll.run(6, out=roll.alloc(6, log_actions=True))      # fine: starts at step 0
    with pytest.raises(hip.lib.MnkHipError):
        roll.run(4, out=roll.alloc(4, log_actions=True))  # would start at step 6
    roll.run(2)                                            # without a log any step is fine
    roll.run(4, out=roll.alloc(4, log_actions=True))      # step 8 again


def test_replay_flags_a_foreign_log(hip):
    from selfplay.random_rollout import GatheredLogs, replay_shard

    env = hip.Env(3, 3, 3, 8, device=DEV)
    logs = GatheredLogs(planes0=env._planes.unsqueeze(0).clone(), meta0=env._meta.unsqueeze(0).clone(),
                        act=torch.full((1, 1, 8), 200, dtype=torch.int32, device=DEV), steps=4)
    err = torch.zeros(2, dtype=torch.int32, device=DEV)
    replay_shard(logs, 0, 3, 3, 3, err=err)
    assert err[0].item() == hip.lib.ERR_ACTION_RANGE
    # the same in the 7-bit stream: action 127 in every field of the first (only) word
    logs7 = GatheredLogs(planes0=env._planes.unsqueeze(0).clone(), meta0=env._meta.unsqueeze(0).clone(),
                         act=torch.full((1, 1, 8), 0x0FFFFFFF, dtype=torch.int32, device=DEV), steps=4, fmt=hip.lib.ACT_BITS7)
    err.zero_()
    replay_shard(logs7, 0, 3, 3, 3, err=err)
    assert err[0].item() == hip.lib.ERR_ACTION_RANGE
    with pytest.raises(ValueError):  # a log-only message cannot be replayed without the receiver's state
        replay_shard(GatheredLogs(planes0=None, meta0=None, act=logs7.act, steps=4, fmt=hip.lib.ACT_BITS7), 0, 3, 3, 3)


def _fuzz_geometries(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        m, n = int(rng.integers(2, 23)), int(rng.integers(2, 23))
        k = int(rng.integers(1, min(m, n) + 1))
        if m * (n + 1) <= 512:
            out.append((m, n, k))
    return out


@pytest.mark.parametrize("m,n,k", _fuzz_geometries(24, seed=2026) + [(22, 22, 10), (9, 9, 4), (9, 9, 9), (2, 22, 2),
                                                                      (22, 2, 2), (15, 15, 5), (12, 9, 5),
                                                                      (1, 7, 1), (1, 61, 1), (5, 5, 1), (3, 61, 3)])  # one-row boards, k = 1, widest row
def test_generic_geometries_match_oracle(hip, m, n, k):
    """Boards outside the ahead-of-time specialisations (and odd ones inside them), both ways they can run: the
    generic kernels with run-time shifts (MNK_JIT=0) -- including shift amounts >= 32 and k up to 10 -- and the
    kernel hiprtc compiles for exactly this board (MNK_JIT=1), against the oracle: a fused rollout (records,
    statistics, final state) and an API-level step with observation and mask, bit for bit."""
    nenv, steps = 67, 3 * max(m, n)
    ora = OracleVectorEnv(m, n, k, nenv)
    planes, meta, stats = random_rollout(ora, seed=m * 1000 + n * 10 + k, step0=0, steps=steps)
    saved = os.environ.get("MNK_JIT")
    try:
        for jit in ("0", "1"):
            os.environ["MNK_JIT"] = jit
            _reload_knobs()
            env = hip.Env(m, n, k, nenv, device=DEV)
            roll = hip.Rollout(env, seed=m * 1000 + n * 10 + k)
            rec = roll.run(steps)
            assert np.array_equal(rec.planes.cpu().numpy().view(np.uint64), planes), f"MNK_JIT={jit}"
            assert np.array_equal(rec.meta.cpu().numpy().view(np.uint32), meta), f"MNK_JIT={jit}"
            assert np.array_equal(roll.stats.cpu().numpy(), stats), f"MNK_JIT={jit}"
            assert torch.equal(env.move_counts.cpu(), ora.move_counts), f"MNK_JIT={jit}"
            # split launches, action log on: the head / tail paths of the specialised kernel
            env2 = hip.Env(m, n, k, nenv, device=DEV)
            roll2 = hip.Rollout(env2, seed=m * 1000 + n * 10 + k)
            first = max(4, (steps // 8) * 4)
            a = roll2.alloc(first, log_actions=True)
            roll2.run(first, out=a)
            b = roll2.run(steps - first)
            assert torch.equal(torch.cat([a.planes, b.planes]), rec.planes) and torch.equal(torch.cat([a.meta, b.meta]), rec.meta)
            assert torch.equal(hip.rollout.unpack_action_log(a.act, first, a.fmt), rec.actions()[:first])
    finally:
        if saved is None:
            os.environ.pop("MNK_JIT", None)
        else:
            os.environ["MNK_JIT"] = saved
        _reload_knobs()
    acts = torch.from_numpy(np.random.default_rng(k).integers(-m * n, m * n, nenv))
    o1, r1, d1 = env.step(acts.to(DEV))
    o2, r2, d2 = ora.step(acts)
    assert torch.equal(o1["observation"].cpu(), o2["observation"]) and torch.equal(o1["action_mask"].cpu(), o2["action_mask"])
    assert torch.equal(r1.cpu(), r2) and torch.equal(d1.cpu(), d2)
    assert torch.equal(env.move_counts.cpu(), ora.move_counts) and torch.equal(env.current_player.cpu(), ora.current_player)


def test_rollout_soak_is_deterministic(hip):
    """2 x 1000 launches of 256 plies on 65 536 envs (3.4e10 env-steps): the same seed gives the same final state
    and counters in two independent runs (determinism check), the counters add up, and the
    long-run statistics sit on the known values of uniform random play."""
    m, n, k, nenv = 9, 9, 5, 65536
    finals = []
    for _ in range(2):
        env = hip.Env(m, n, k, nenv, device=DEV)
        roll = hip.Rollout(env, seed=123)
        buf = roll.alloc(256)
        for _ in range(1000):
            roll.run(256, out=buf)
        finals.append((env._planes.clone(), env._meta.clone(), roll.stats.clone(), buf.meta.clone()))
    for a, b in zip(*finals):
        assert torch.equal(a, b)
    episodes, black, white, draws, length = finals[0][2].tolist()
    assert episodes == black + white + draws
    # unbiased reference values: first-game statistics of 327 680 games of the imported reference (its env and its
    # multinomial RandomPolicy), tests/golden/random_play_stats.npz made by tests/golden/make_golden_stats.py
    # (BASELINE.md's 53.3 / 0.26 % came from a fixed window after a common start, which under-counts long games).
    # This run has ~6e8 games, so the tolerance is the fixture's own standard error (x 4).
    ref = random_play_stats("9x9x5")
    assert abs(length / episodes - ref["mean_plies"]) < 4 * ref["se_mean_plies"], (length / episodes, ref)
    assert abs(draws / episodes - ref["draw_rate"]) < 4 * ref["se_draw_rate"], (draws / episodes, ref)
    share = ref["black_wins"] / ref["games"]
    assert abs(black / episodes - share) < 4 * (share * (1 - share) / ref["games"]) ** 0.5  # the first-move advantage


def test_oversized_emit_threads_setting_is_refused_not_launched():
    """Round-1 fault: MNK_EMIT_THREADS=512 launched 512 threads into __launch_bounds__(256) kernels
    ("unspecified launch failure").  mnk_block_threads() now falls back to 256 for anything but 64/128/256; the
    value is cached per process, so the check runs in a fresh child process: observe + step on a small env must
    equal the oracle with the bad setting in the environment."""
    import subprocess
    import sys

    child = r"""
import os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "rl-selfplay-mnk_amd"), os.path.join(%(root)r, "tests")]
import numpy as np, torch
from env.torch_vector_mnk_env import TorchVectorMnkEnv
from oracle.env_torch import OracleVectorEnv
env, ora = TorchVectorMnkEnv(9, 9, 5, 200, device="cuda:0"), OracleVectorEnv(9, 9, 5, 200)
rng = np.random.default_rng(0)
for t in range(6):
    a = torch.from_numpy(rng.integers(0, 81, 200))
    o1, r1, d1 = env.step(a.to("cuda:0"))
    o2, r2, d2 = ora.step(a)
    assert torch.equal(o1["observation"].cpu(), o2["observation"]) and torch.equal(o1["action_mask"].cpu(), o2["action_mask"])
    assert torch.equal(r1.cpu(), r2) and torch.equal(d1.cpu(), d2)
o1, o2 = env.observe(), ora.observe()
assert torch.equal(o1["observation"].cpu(), o2["observation"])
torch.cuda.synchronize()
print("EMIT_OK")
""" % {"root": os.path.dirname(os.path.dirname(os.path.abspath(__file__)))}
    env = dict(os.environ, MNK_EMIT_THREADS="512", MNK_EMIT_ENVS="48")
    out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=300)
    assert "EMIT_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    # round-3 advisor finding: MNK_EMIT_ENVS=128 is legal and so is MNK_EMIT_THREADS=64, but together the envs 64..127 of
    # every workgroup had no lane to play them.  mnk_block_threads() now never hands out fewer threads than envs.
    for knobs in ({"MNK_EMIT_ENVS": "128", "MNK_EMIT_THREADS": "64"}, {"MNK_EMIT_ENVS": "128"}, {"MNK_EMIT_ENVS": "16", "MNK_EMIT_THREADS": "64"}):
        out = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, **knobs), capture_output=True, text=True, timeout=300)
        assert "EMIT_OK" in out.stdout, str(knobs) + out.stdout[-2000:] + out.stderr[-4000:]


def test_empty_and_single_env_batches(hip):
    """Edge sizes: an env with no envs at all is a no-op everywhere (the reference's tensors simply have a zero
    dimension), and one env works like any other batch."""
    from selfplay.policy import RandomPolicy
    from selfplay.torch_self_play_wrapper import TorchSelfPlayWrapper

    env = hip.Env(9, 9, 5, 0, device=DEV)
    obs = env.reset()
    assert obs["observation"].shape == (0, 2, 9, 9) and obs["action_mask"].shape == (0, 81)
    o, r, d = env.step(torch.zeros(0, dtype=torch.long, device=DEV))
    assert r.shape == (0,) and d.shape == (0,) and o["action_mask"].dtype == torch.bool
    o, r, d = env.step_subset(torch.zeros(0, dtype=torch.long, device=DEV), torch.zeros(0, dtype=torch.long, device=DEV))
    assert r.shape == (0,)
    env.reset(torch.zeros(0, dtype=torch.long, device=DEV))
    rec = hip.Rollout(env, seed=1).run(16)
    assert rec.planes.shape == (16, 3, 0) and rec.meta.shape == (16, 0)
    wrap = TorchSelfPlayWrapper(env, seed=1)
    wrap.set_opponent(RandomPolicy(81, seed=2))
    o, _ = wrap.reset()
    o, r, t, tr, _ = wrap.step(torch.zeros(0, dtype=torch.long, device=DEV))
    assert r.shape == (0,) and t.shape == (0,) and tr.shape == (0,)
    assert RandomPolicy(81, seed=3).act(o).shape == (0,)
    env.check_errors()

    one, ora = hip.Env(9, 9, 5, 1, device=DEV), OracleVectorEnv(9, 9, 5, 1)
    rng = np.random.default_rng(5)
    for _ in range(40):
        a = torch.from_numpy(rng.integers(0, 81, 1))
        o1, r1, d1 = one.step(a.to(DEV))
        o2, r2, d2 = ora.step(a)
        assert torch.equal(o1["observation"].cpu(), o2["observation"]) and torch.equal(r1.cpu(), r2) and torch.equal(d1.cpu(), d2)
        if bool(d2.any()):
            one.reset(torch.tensor([0], device=DEV))
            ora.reset(torch.tensor([0]))


@pytest.mark.parametrize("m,n,k,nenv", [(3, 3, 3, 300), (9, 9, 5, 130), (7, 9, 7, 64)])
def test_step_with_autoreset_equals_step_reset_observe(hip, m, n, k, nenv):
    """MNK_STEP_AUTORESET: one launch == the raw loop's step(a); reset(nonzero(done)); observe() on the oracle
    (env:55-84, :34-44, :46-53): rewards / dones of the ply, state and mask / observation of the restarted games."""
    env, ora = hip.Env(m, n, k, nenv, device=DEV), OracleVectorEnv(m, n, k, nenv)
    rng = np.random.default_rng(m * n)
    acts = torch.empty(nenv, dtype=torch.long, device=DEV)
    rew = torch.empty(nenv, dtype=torch.float32, device=DEV)
    done = torch.empty(nenv, dtype=torch.bool, device=DEV)
    mask = torch.empty((nenv, m * n), dtype=torch.bool, device=DEV)
    obs = torch.empty((nenv, 2, m, n), dtype=torch.float32, device=DEV)
    finished = 0
    want = ora.observe()
    for t in range(4 * m * n):
        legal = want["action_mask"].numpy()
        a = np.array([rng.choice(np.nonzero(row)[0]) for row in legal])
        acts.copy_(torch.from_numpy(a))
        env.step_into(acts, rew, done, mask, obs, autoreset=True)
        _, r2, d2 = ora.step(torch.from_numpy(a))
        if bool(d2.any()):
            ora.reset(torch.nonzero(d2).squeeze(1))
        want = ora.observe()
        finished += int(d2.sum())
        assert torch.equal(rew.cpu(), r2) and torch.equal(done.cpu(), d2), t
        assert torch.equal(mask.cpu(), want["action_mask"]) and torch.equal(obs.cpu(), want["observation"]), t
        assert torch.equal(env.move_counts.cpu(), ora.move_counts) and torch.equal(env.current_player.cpu(), ora.current_player)
    assert finished > nenv
    with pytest.raises(hip.lib.MnkHipError):  # subset steps have no autoreset form
        idx = torch.arange(4, device=DEV)
        hip.lib.call("mnk_step", hip.lib.ptr(env._planes), hip.lib.ptr(env._meta), nenv, m, n, k, hip.lib.ptr(acts[:4].contiguous()),
                     hip.lib.ptr(idx), 4, hip.lib.ptr(rew), hip.lib.ptr(done), None, None, hip.lib.OBS_F32,
                     hip.lib.ptr(env._err), hip.lib.STEP_AUTORESET, env._stream())


def test_long_launch_falls_back_to_64_bit_record_addresses(hip):
    """Up to 65 536 envs the record stores use 32-bit byte offsets, valid while one launch's record rows stay below
    4 GiB (9x9x5 x 65 536 envs: 2 730 plies).  A longer launch must take the 64-bit form and give the same records
    as two launches of half the length (which take the 32-bit form)."""
    m, n, k, nenv, steps = 9, 9, 5, 65536, 3000
    a = hip.Rollout(hip.Env(m, n, k, nenv, device=DEV), seed=21)
    whole = a.run(steps)
    assert whole.planes.numel() * 8 > (1 << 32)
    b = hip.Rollout(hip.Env(m, n, k, nenv, device=DEV), seed=21)
    first = b.run(steps // 2)
    assert torch.equal(first.planes, whole.planes[:steps // 2]) and torch.equal(first.meta, whole.meta[:steps // 2])
    del first
    second = b.run(steps - steps // 2)
    assert torch.equal(second.planes, whole.planes[steps // 2:]) and torch.equal(second.meta, whole.meta[steps // 2:])
    assert torch.equal(a.stats, b.stats) and torch.equal(a.env._planes, b.env._planes)
