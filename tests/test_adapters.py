"""Adapters: classic gym (pyprojectd/projectd_gym/projectd_gym.py:10-42, __init__.py:1-8), gymnasium
(projectd_gymnasium/projectd_gymnasium.py:9-40) and the SB3-shaped VecEnv with sac.yml's observation normalisation.  gym /
gymnasium / stable_baselines3 are not installed here: the modules run against tests/stubs (see its README) -- enough to execute
every line of the adapters; construction and spaces on the CPU, stepping under -m gpu."""
import importlib, os, sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def stubs():
    used = {}
    for name in ('gym', 'gymnasium'):
        try:
            m = importlib.import_module(name)
        except ImportError:
            if os.path.join(HERE, 'stubs') not in sys.path:
                sys.path.append(os.path.join(HERE, 'stubs'))
            m = importlib.import_module(name)
        used[name] = 'tests/stubs stand-in' if os.path.join(HERE, 'stubs') in os.path.abspath(getattr(m, '__file__', '')) else 'the installed package %s' % getattr(m, '__version__', '')
    try:
        import stable_baselines3 as _sb3
        used['stable_baselines3'] = 'the installed package %s' % _sb3.__version__
    except ImportError:
        used['stable_baselines3'] = 'absent (projectd_sb3 is exercised through its own VecEnv-shaped surface)'
    print('adapter tests run against: ' + '; '.join('%s = %s' % kv for kv in used.items()))   # (pytest -s / the report's captured output says which)
    return used


@pytest.fixture(scope='module')
def base(built):
    import tempfile, synthetic_tracks
    d = tempfile.mkdtemp(prefix='pdb_adapters_')
    synthetic_tracks.make_base(d, tracks=('flat', 'touge'))
    synthetic_tracks.install_packed_car(d)
    os.environ['PROJECTD_BASE'] = d
    return d


def _check_spaces(env):
    assert env.observation_space.shape == (24,) and env.action_space.shape == (2,)
    assert env.observation_space.low[6] == 0.0 and env.observation_space.high[0] == 100.0 and env.observation_space.high[17] == 50.0
    assert list(env.action_space.low) == [-1.0, -1.0] and list(env.action_space.high) == [1.0, 1.0]


def test_classic_gym_adapter_constructs_and_registers(stubs, base):
    import gym, projectd_gym
    projectd_gym.register()
    spec = gym.envs.registration.registry['ProjectD-v0']
    assert (spec['max_episode_steps'] if isinstance(spec, dict) else spec.max_episode_steps) == 80000
    env = projectd_gym.ProjectDEnvGym(track_name='flat')
    _check_spaces(env)
    assert isinstance(env, gym.Env) and isinstance(env.seed(7), list)
    env.close()


def test_gymnasium_adapter_constructs_and_registers(stubs, base):
    import gymnasium, projectd_gymnasium
    projectd_gymnasium.register()
    spec = gymnasium.envs.registration.registry['ProjectD-v0']
    assert (spec['max_episode_steps'] if isinstance(spec, dict) else spec.max_episode_steps) == 80000
    env = projectd_gymnasium.ProjectDGymnasium(track_name='flat')
    _check_spaces(env)
    assert isinstance(env, gymnasium.Env)
    env.seed(3)
    env.close()


def test_running_obs_norm_matches_the_batch_statistics():
    """RunningObsNorm = SB3 RunningMeanStd: after any sequence of batches, mean / var equal those of everything seen"""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'projectd-core_amd'))

    class Fake:
        num_envs = 8
        class observation_space: shape = (5,)
        action_space = None
        def __init__(self): self.rng = np.random.RandomState(0); self.seen = []
        def reset(self): x = self.rng.normal(3, 2, (8, 5)).astype(np.float32); self.seen.append(x); return x
        def step(self, a):
            x = self.rng.normal(-1, 5, (8, 5)).astype(np.float32); self.seen.append(x)
            done = np.zeros(8, bool); done[2] = True
            return x, np.zeros(8, np.float32), done, [{'terminal_observation': x[i].copy()} if done[i] else {} for i in range(8)]
        def close(self): pass
    try:
        import projectd_sb3
    except ImportError:
        pytest.skip('PyProjectD module not built')
    f = Fake()
    n = projectd_sb3.RunningObsNorm(f)
    n.reset()
    for _ in range(20):
        o, r, d, infos = n.step(None)
    allx = np.concatenate(f.seen).astype(np.float64)
    assert np.allclose(n.mean, allx.mean(0), atol=1e-3) and np.allclose(n.var, allx.var(0), rtol=1e-3)
    assert o.dtype == np.float32 and np.abs(o).max() <= 10.0 and 'terminal_observation' in infos[2]
    n.training = False
    m = n.mean.copy(); n.step(None)
    assert np.array_equal(m, n.mean)


@pytest.mark.gpu
def test_gym_and_gymnasium_envs_step_like_the_plain_env(stubs, base):
    import gym, gymnasium, projectd_gym, projectd_gymnasium, projectd_env as E
    projectd_gym.register(); projectd_gymnasium.register()
    g = gym.make('ProjectD-v0', track_name='flat')
    gn = gymnasium.make('ProjectD-v0', track_name='flat')
    plain = E.ProjectDEnv(track_name='flat')
    o1 = g.reset(); o2, info = gn.reset(seed=1); o3 = plain.reset()
    assert np.array_equal(o1, o3) and np.array_equal(o2, o3) and info == {}
    rng = np.random.RandomState(5)
    for t in range(120):
        a = rng.uniform(-1, 1, 2).astype(np.float32)
        s1 = g.step(a); s2 = gn.step(a); s3 = plain.step(a)
        assert len(s1) == 4 and len(s2) == 5
        assert np.array_equal(s1[0], s3[0]) and np.array_equal(s2[0], s3[0])
        assert s1[1] == s3[1] and s2[1] == s3[1] and s1[2] == (s3[2] or s3[3]) and s2[2] == s3[2]
        assert g.observation_space.contains(s1[0])
    g.close(); gn.close(); plain.close()


@pytest.mark.gpu
def test_sb3_vec_env_episode_protocol(stubs, base):
    """next-step form (same_step_reset=False): done / terminal_observation / reset_tick / TimeLimit.truncated on the batch, with the observation normaliser on top"""
    import projectd_sb3
    n = 16
    env = projectd_sb3.ProjectDSB3VecEnv(n, track_name='flat', stuck_timeout=0.3, max_episode_steps=100000, terminate_low_reward=-1e9, same_step_reset=False)
    venv = projectd_sb3.wrap_normalize(env)
    obs = venv.reset()
    assert obs.shape == (n, 24) and obs.dtype == np.float32
    acts = np.zeros((n, 2), np.float32); acts[:, 1] = -1.0       # creeping: no new track point within 0.3 s => the stuck rule ends the episode
    dones = np.zeros(n, int); resets = np.zeros(n, int); trunc = np.zeros(n, int)
    pending = np.zeros(n, bool)
    for t in range(500):
        if t == 250:
            env.max_episode_steps = 60                             # from here the time limit comes first
        obs, rew, done, infos = venv.step(acts)
        assert obs.shape == (n, 24) and np.abs(obs).max() <= 10.0
        for i in range(n):
            if done[i]:
                assert 'terminal_observation' in infos[i] and infos[i]['terminal_observation'].shape == (24,)
                trunc[i] += int(infos[i]['TimeLimit.truncated'])
                if t < 250:
                    assert not infos[i]['TimeLimit.truncated']
            if pending[i]:
                assert infos[i].get('reset_tick') is True and rew[i] == 0.0 and not done[i]
                resets[i] += 1
        dones += done; pending = done.copy()
    assert dones.min() >= 4 and trunc.min() >= 2, (dones, trunc)
    assert np.array_equal(resets, dones) or np.array_equal(resets + pending, dones)
    assert env.env_is_wrapped(object) == [False] * n and env.get_attr('track_name') == ['flat'] * n
    env.seed(11)
    venv.close()


def _lane_actions(lane, k):
    """the action lane `lane` gives on its k-th real (non-reset) step"""
    return np.array([0.35 * np.sin(0.7 * lane + 0.013 * k), 0.2 + 0.7 * np.cos(0.31 * lane + 0.004 * k)], np.float32)


@pytest.mark.gpu
def test_same_step_reset_is_the_next_step_stream_with_the_reset_ticks_folded_in(stubs, base):
    """stable-baselines3's convention (the SB3 adapter's default, pdb_step_host_held): on the step where a lane's episode ends the observation returned
    is the NEW episode's first one, the terminal one goes to infos[i]['terminal_observation'], and no lane ever spends a step on a reset tick.  Lane by
    lane the stream equals the next-step env's (itself held against the oracle, tests/test_pyprojectd_api.py) with every reset-tick transition folded
    into the step before it -- observations, rewards, dones, bit for bit -- while the lanes that did not end an episode are untouched by the extra launch.
    Episodes end by the stuck rule, by leaving the road and by the time limit."""
    got_done = np.zeros(12, int); got_trunc = np.zeros(12, int)
    for limit in (140, 60):     # the stuck rule fires after 0.4 s = 133 steps: first it ends the episodes, then the time limit does
        d1, t1 = _same_step_against_next_step(12, dict(track_name='flat', stuck_timeout=0.4, terminate_low_reward=-1e9, max_episode_steps=limit))
        got_done += d1; got_trunc += t1
    assert got_done.min() >= 6 and got_trunc.min() >= 3 and (got_done - got_trunc).min() >= 3, (got_done, got_trunc)


def _same_step_against_next_step(n, kw):
    import projectd_sb3
    A = projectd_sb3.ProjectDSB3VecEnv(n, same_step_reset=False, **kw)     # next-step: the reference stream
    B = projectd_sb3.ProjectDSB3VecEnv(n, same_step_reset=True, **kw)
    oa = A.reset(); ob = B.reset()
    assert np.array_equal(oa, ob)
    ka = np.zeros(n, int); kb = np.zeros(n, int)                 # real steps taken per lane
    # per lane: A's transitions with the reset ticks folded in, as a queue B's are checked against
    want = [[] for _ in range(n)]
    open_done = [None] * n                                        # A: a done transition waiting for its reset tick's observation
    got_done = np.zeros(n, int); got_trunc = np.zeros(n, int)
    slow = np.arange(n) % 3 == 0                                  # these lanes creep (stuck rule); the others drive, some off the road
    for t in range(700):
        acts = np.stack([_lane_actions(i, ka[i]) for i in range(n)]); acts[slow, 1] = -1.0
        o, r, d, infos = A.step(acts)
        for i in range(n):
            if infos[i].get('reset_tick'):                        # the reset tick: its observation completes the transition before it
                assert open_done[i] is not None and r[i] == 0.0 and not d[i]
                tr = open_done[i]; open_done[i] = None
                want[i].append((o[i].copy(), tr[1], True, tr[0], tr[2]))
                continue
            ka[i] += 1
            if d[i]:
                assert np.array_equal(infos[i]['terminal_observation'], o[i])
                open_done[i] = (o[i].copy(), r[i], bool(infos[i]['TimeLimit.truncated']))
            else:
                want[i].append((o[i].copy(), r[i], False, None, False))
    for t in range(560):
        acts = np.stack([_lane_actions(i, kb[i]) for i in range(n)]); acts[slow, 1] = -1.0
        o, r, d, infos = B.step(acts)
        kb += 1
        for i in range(n):
            assert 'reset_tick' not in infos[i]
            wo, wr, wd, wterm, wtrunc = want[i].pop(0)
            assert d[i] == wd and r[i] == wr and np.array_equal(o[i], wo), (t, i)
            if d[i]:
                assert np.array_equal(infos[i]['terminal_observation'], wterm) and infos[i]['TimeLimit.truncated'] == wtrunc
                got_done[i] += 1; got_trunc[i] += int(wtrunc)
    A.close(); B.close()
    return got_done, got_trunc
