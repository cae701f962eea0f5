"""GPU tests of VecBenchmarkPushingEnv.

What the reference pins is tested against fixtures / the reference's rules; the contact physics (engine model gpr-push-v3,
parity unpinned -- MuJoCo is not available) is tested for HIP == oracle, for invariants any rigid-body contact model must keep
(momentum under frictionless contact, no energy gain) and for physical sanity."""
import numpy as np
import pytest

from conftest import load_npz, record_metric

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gpr():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import gymnasium_planar_robotics_amd as g
    g._native.load()
    return g


def _g(t):
    return t.double().cpu().numpy() if t.dtype in (torch.float32, torch.float64) else t.cpu().numpy()


@pytest.mark.parametrize('precision,tol', [('f64', 1e-12), ('f32', 5e-5)])
def test_free_space_actuation_matches_reference_closed_form(gpr, precision, tol):
    """the 24 parametrisations of the reference's tests/test_benchmark_pushing_env.py (object far away: no contact)"""
    z = load_npz('trajectories.npz')
    n = 0
    for i in range(int(z['n'])):
        if str(z[f'{i}_env']) != 'pushing':
            continue
        mode = str(z[f'{i}_mode']); ncyc = int(z[f'{i}_num_cycles'])
        env = gpr.VecBenchmarkPushingEnv(std_noise=0.0, num_cycles=ncyc, v_max=0.01, a_max=0.2, j_max=150.0, learn_jerk=(mode == 'jerk'),
                                         max_episode_steps=None, autoreset_mode='disabled', precision=precision, num_envs=1)
        env.reset(seed=0)
        env.pos[0] = torch.as_tensor(z[f'{i}_starts'].reshape(-1), dtype=env.dtype)   # (0.48, 0.48), PU test :38
        env.object[0, :2] = torch.as_tensor([0.12, 0.36], dtype=env.dtype)              # PU:176
        act = torch.as_tensor(z[f'{i}_deriv'][None, :], dtype=env.dtype, device=env.device)
        pos, vel = [], []
        for _ in range(100):
            env.step(act)
            pos.append(_g(env.pos[0])); vel.append(_g(env.vel[0]))
        assert np.abs(np.array(pos) - z[f'{i}_pos']).max() < tol and np.abs(np.array(vel) - z[f'{i}_vel']).max() < tol, i
        assert float(env.object[0, 3:].abs().max()) == 0.0  # the object was never touched
        env.close(); n += 1
    assert n == 24


@pytest.mark.parametrize('precision', ['f64', 'f32_f64acc', 'f32'])
@pytest.mark.parametrize('kw', [{}, {'learn_jerk': True}, {'autoreset_mode': 'next_step', 'std_noise': 0.0}, {'num_movers': 2},
                                {'num_movers': 2, 'learn_jerk': True, 'E': 4096}, {'num_movers': 2, 'pair_lanes': '0'}])
def test_step_reset_vs_oracle(gpr, oracle, precision, kw, monkeypatch):
    kw = dict(kw)
    E = kw.pop('E', 333)
    NM = kw.get('num_movers', 1)
    if E > 1000 and precision != 'f64':
        pytest.skip('the 4 096-env run of BASELINE config 3 is compared in the pinning precision only')
    # two movers in float: a batch this small runs one lane per (mover, object) pair (k_push<..., PAIR>); the form with both pairs in every
    # lane -- what batches beyond two wavefronts per SIMD run -- is held to the same bounds by switching pair lanes off
    pair = kw.pop('pair_lanes', None)
    if pair is not None:
        if precision != 'f32':
            pytest.skip('pair lanes exist in the float build only')
        monkeypatch.setenv('GPR_PUSH_PAIR', pair)
    autoreset = kw.pop('autoreset_mode', 'same_step')
    genv = gpr.VecBenchmarkPushingEnv(num_envs=E, precision=precision, autoreset_mode=autoreset, env_id_offset=77, **kw)
    okw = dict(kw); okw.setdefault('std_noise', 1e-5)
    oenv = oracle.OraclePushingEnv(E, autoreset=autoreset, storage_f32=(precision != 'f64'), env_id_offset=77, **okw)
    tol = {'f64': 1e-9, 'f32_f64acc': 3e-6, 'f32': 1e-4}[precision]
    g0 = genv.reset(seed=3); o0 = oenv.reset(seed=3)
    assert np.abs(_g(g0[0]['achieved_goal']) - o0['achieved_goal']).max() < tol and np.abs(_g(genv.goals) - oenv.goal).max() < tol
    for i in range(NM):
        assert (np.linalg.norm(oenv.object[:, :2] - oenv.pos[:, 2 * i:2 * i + 2], axis=1) > genv.min_mo_dist).mean() > 0.99  # PU:407
    assert np.abs(_g(genv.pos) - oenv.pos).max() < tol and np.abs(_g(genv.object) - oenv.object).max() < tol
    rng = np.random.default_rng(1)
    lim = genv.j_max if genv.learn_jerk else genv.a_max
    n_contact = n_term = mism = n_events = 0
    worst = 0.0
    for step in range(70):
        # steer towards the object most of the time so that pushing actually happens
        a = np.zeros((E, 2 * NM))
        for i in range(NM):
            to_obj = oenv.object[:, :2] - oenv.pos[:, 2 * i:2 * i + 2]
            a[:, 2 * i:2 * i + 2] = np.where(rng.random((E, 1)) < 0.7, to_obj / (np.linalg.norm(to_obj, axis=1, keepdims=True) + 1e-9) * lim * 0.6,
                                             rng.uniform(-lim, lim, size=(E, 2)))
        a = a.astype(np.float32).astype(np.float64)
        dev, dt = genv.device, genv.dtype
        for gt, ot in ((genv.pos, oenv.pos), (genv.vel, oenv.vel), (genv.acc, oenv.acc), (genv.goals, oenv.goal), (genv.mover_rot, oenv.rot),
                       (genv.object, oenv.object)):
            gt.copy_(torch.as_tensor(ot, dtype=dt, device=dev))
        genv.elapsed_steps.copy_(torch.as_tensor(oenv.elapsed, device=dev))
        obs, rew, term, trunc, info = genv.step(torch.as_tensor(a, dtype=dt, device=dev))
        o = oenv.step(a)
        safe = o['margin'] > {'f64': 1e-12, 'f32_f64acc': 3e-7, 'f32': 1e-5}[precision]
        agree = ((_g(term) == o['terminated']) & (_g(trunc) == o['truncated']) & (_g(info['is_success']) == o['is_success'])
                 & (_g(info['mover_collision']) == o['mover_collision']))
        bad = ~(agree | ~safe)
        assert not bad.any(), (step, int(bad.sum()), _g(term)[bad], o['terminated'][bad], _g(info['is_success'])[bad], o['is_success'][bad], _g(info['mover_collision'])[bad],
                               o['mover_collision'][bad], o['margin'][bad], np.abs(_g(genv.object) - oenv.object)[bad], np.abs(_g(genv.vel) - oenv.vel)[bad])
        mism += int((~agree).sum())
        ok = agree & safe  # (a wall hit found one cycle apart ends the step one cycle apart: compare states where decisions are safe)
        oerr = np.abs(_g(genv.object) - oenv.object)
        if precision != 'f64':
            # The float builds may see a contact point enter / leave the manifold one cycle earlier or later than the fp64 oracle
            # (stiff impulses: a 10 g object against a 1.24 kg mover): a discrete event, bounded here in frequency and in size.
            # The fp64 build has no such events and pins the algorithm at 1e-9.
            # (with two movers the same event shows in the velocity of the mover that is in contact). An event is a deviation
            # beyond the tolerance the non-events are then held to: scratch/push_err.py measures, over 137 000 env-steps of this
            # workload, mover velocities off by more than 1e-4 in 0.005 % (one mover) / 0.008 % (two movers) of the env-steps (round 2:
            # 0.011 % / 0.026 %, before the float build integrated displacements) and object velocities off by more than 5e-3 in 0.04 % /
            # 0.014 % (0.085 % / 0.092 %) -- a continuum of event sizes, not two classes with a gap between them.
            verr = np.abs(_g(genv.vel) - oenv.vel).max(axis=1)
            event = (oerr[:, 3:5].max(axis=1) > 50 * tol) | (verr > tol)
            assert event.mean() < 0.01 and oerr[:, 3:5].max() < 0.05 and oerr[:, :2].max() < 2e-3, (step, event.mean(), oerr.max(axis=0))
            assert np.abs(_g(obs['observation'])[ok] - o['observation'][ok]).max() < 5e-3, step
            worst = max(worst, float(oerr[:, 3:5].max()))
            n_events += int(event[agree & safe].sum())
            ok = ok & ~event
        assert np.abs(_g(obs['observation'])[ok] - o['observation'][ok]).max() < tol, step
        assert np.abs(_g(obs['achieved_goal'])[ok] - o['achieved_goal'][ok]).max() < (tol if precision == 'f64' else 5 * tol), step
        assert np.array_equal(_g(rew)[agree], o['reward'][agree])
        assert oerr[ok][:, :2].max() < (tol if precision == 'f64' else 5 * tol), step
        assert oerr[ok][:, 3:5].max() < (tol if precision == 'f64' else 50 * tol), step
        n_contact += int((np.abs(oenv.object[:, 3:5]).max(axis=1) > 0).sum()); n_term += int(o['terminated'].sum())
    assert n_contact > 100 and n_term > 0, (n_contact, n_term)
    assert mism <= (0 if precision == 'f64' else 5)
    # gpr-push-v3 (round 4): scratch/push_err.py over 137 000 env-steps of this workload -- observations off by > 1e-4 in 0 % (one mover) /
    # 0.003 % (two movers) of the env-steps, object velocities off by > 1e-4 m/s in 0.01 % (v2: 0.005-0.008 % / 0.04 % beyond 50x that)
    assert n_events <= max(3, 5e-4 * 70 * E), (n_events, 70 * E)
    record_metric(f'pushing_step_reset_vs_oracle[{precision}-NM{NM}-E{E}-{"jerk" if kw.get("learn_jerk") else "acc"}-{autoreset}{"-one-lane" if pair == "0" else ""}]',
                  flag_mismatches=mism, env_steps=70 * E, worst_object_velocity_event=worst, events=n_events, tol=tol)
    genv.close()


def test_reset_options_fix_mover_start_and_goal(gpr, oracle):
    """the reference's reset options (PU:389-391, 412-414): given mover start(s) / goal are taken as they are, the object's start is
    still rejection-sampled to keep min_mo_dist from the given start (PU:396-407); identical to the oracle's samples"""
    for NM in (1, 2):
        E = 512
        genv = gpr.VecBenchmarkPushingEnv(num_envs=E, precision='f64', num_movers=NM, std_noise=0.0)
        oenv = oracle.OraclePushingEnv(E, num_movers=NM, std_noise=0.0)
        start = np.array([[0.2, 0.5], [0.55, 0.2]])[:NM]
        goal = np.array([0.4, 0.3])
        obs, info = genv.reset(seed=5, options={'mover_start_xy_pos': start, 'object_goal_xy_pos': goal})
        o = oenv.reset(seed=5, mover_start_xy_pos=start, object_goal_xy_pos=goal)
        assert np.array_equal(_g(genv.pos), np.tile(start.reshape(1, -1), (E, 1))) and np.array_equal(_g(genv.goals), np.tile(goal, (E, 1)))
        assert np.array_equal(_g(obs['desired_goal']), np.tile(goal, (E, 1)))
        assert np.array_equal(_g(genv.object), oenv.object) and np.array_equal(_g(obs['achieved_goal']), o['achieved_goal'])
        for i in range(NM):
            assert (np.linalg.norm(_g(genv.object)[:, :2] - start[i], axis=1) > genv.min_mo_dist).all()
        assert len(np.unique(_g(genv.object)[:, 0])) > E // 2  # the object start is still random
        # per-env values, and only one of the two options
        per_env = np.stack([np.linspace(0.15, 0.55, E), np.full(E, 0.3)], axis=1)
        if NM == 2:
            per_env = np.concatenate([per_env, np.stack([np.full(E, 0.3), np.full(E, 0.58)], axis=1)], axis=1)
        genv.reset(options={'mover_start_xy_pos': per_env.reshape(E, NM, 2)}); oenv.reset(mover_start_xy_pos=per_env)
        assert np.array_equal(_g(genv.pos), per_env) and np.array_equal(_g(genv.object), oenv.object) and np.array_equal(_g(genv.goals), oenv.goal)
        assert len(np.unique(_g(genv.goals)[:, 0])) > E // 2  # the goal was sampled this time
        genv.close()


@pytest.mark.parametrize('precision,tol', [('f64', 1e-12), ('f32', 2e-5)])
@pytest.mark.parametrize('NM', [1, 2])
def test_contact_model_invariants(gpr, precision, tol, NM):
    """What an unpinned contact model can still be held to: with friction, damping and the yaw impedance switched off the impulses
    between mover(s) and object are the only forces, so total linear momentum is conserved through every contact; and contact never
    creates kinetic energy (the impulses are inelastic, the penetration recovery is a velocity bias, not a spring)."""
    E = 2048
    ov = dict(friction_tile=0.0, friction_contact=0.0, friction_torsional=0.0, object_damping=0.0, rot_stiffness=0.0)
    env = gpr.VecBenchmarkPushingEnv(num_envs=E, precision=precision, num_movers=NM, std_noise=0.0, autoreset_mode='disabled',
                                     max_episode_steps=None, _push_overrides=ov)
    env._push_params.object_noise = 0.0
    env.reset(seed=1)
    g = torch.Generator(device='cuda').manual_seed(0)
    U = lambda lo, hi, *s: torch.rand(*s, device='cuda', generator=g, dtype=torch.float64) * (hi - lo) + lo  # noqa: E731
    env.object.zero_(); env.object[:, 0] = 0.36; env.object[:, 1] = 0.36; env.object[:, 2] = U(-0.5, 0.5, E).to(env.dtype)
    ang = U(0, 2 * np.pi, E, NM)
    if NM == 2:
        ang[:, 1] = ang[:, 0] + np.pi + U(-0.5, 0.5, E)
    sp, off = U(0.05, 0.3, E, NM), U(-0.3, 0.3, E, NM)
    pos = torch.stack([0.36 + 0.16 * torch.cos(ang), 0.36 + 0.16 * torch.sin(ang)], dim=-1).reshape(E, 2 * NM)
    vel = torch.stack([-sp * torch.cos(ang + off), -sp * torch.sin(ang + off)], dim=-1).reshape(E, 2 * NM)
    env.pos.copy_(pos.to(env.dtype)); env.vel.copy_(vel.to(env.dtype))
    mm, mo, ax, h = 1.24, 0.01, 0.0775, 0.035
    Im, Io = mm * (2 * ax * ax) / 3, mo * (2 * h * h) / 3

    def momentum():
        return mo * env.object[:, 3:5].double() + mm * env.vel.double().reshape(E, NM, 2).sum(dim=1)

    def energy():
        k = 0.5 * mo * (env.object[:, 3:5].double() ** 2).sum(dim=1) + 0.5 * Io * env.object[:, 5].double() ** 2
        return k + 0.5 * mm * (env.vel.double() ** 2).sum(dim=1) + 0.5 * Im * (env.mover_rot.double().reshape(E, NM, 2)[:, :, 1] ** 2).sum(dim=1)

    p0, k0 = momentum(), energy()
    touched = torch.zeros(E, dtype=torch.bool, device='cuda')
    alive = torch.ones(E, dtype=torch.bool, device='cuda')
    worst_p = worst_k = 0.0
    for t in range(6):
        obs, rew, term, trunc, info = env.step(torch.zeros(E, 2 * NM, dtype=env.dtype, device='cuda'))
        alive &= ~(info['wall_collision'] | info['mover_collision'])  # (an env that crashed stops mid-step)
        touched |= env.object[:, 3:5].abs().max(dim=1).values > 0
        worst_p = max(worst_p, float((momentum() - p0)[alive].abs().max()))
        worst_k = max(worst_k, float(((energy() - k0) / k0)[alive].max()))
    assert touched.float().mean() > 0.6 and alive.float().mean() > 0.5
    assert worst_p < tol * 10 and worst_k < tol * 100, (worst_p, worst_k)
    record_metric(f'pushing_invariants[{precision}-NM{NM}]', momentum_drift=worst_p, relative_energy_gain=worst_k, tol=tol)
    env.close()


def test_pushing_semantics_and_physical_sanity(gpr):
    E = 4096
    env = gpr.VecBenchmarkPushingEnv(num_envs=E, precision='f32', std_noise=0.0)
    obs, info = env.reset(seed=9)
    assert obs['observation'].shape == (E, 4) and obs['achieved_goal'].shape == (E, 2)
    lo, hi = torch.as_tensor(env.object_min_xy_pos, device='cuda'), torch.as_tensor(env.object_max_xy_pos, device='cuda')
    assert (env.object[:, :2] >= lo - 1e-6).all() and (env.object[:, :2] <= hi + 1e-6).all() and (env.goals >= lo - 1e-6).all()
    # PU:396-407 rejects object positions closer than min_mo_dist to the mover. Where the mover starts near the centre almost no
    # object position qualifies (the reference then loops for a very long time); the engine stops after max_reset_trials.
    assert ((env.object[:, :2] - env.pos).norm(dim=1) <= env.min_mo_dist - 1e-6).float().mean() < 0.005
    # untouched objects never move; pushed objects move in the direction they are pushed; nothing tunnels
    start = env.object[:, :2].clone()
    touched = torch.zeros(E, dtype=torch.bool, device='cuda')
    for step in range(30):
        d = env.object[:, :2] - env.pos
        obs, rew, term, trunc, info = env.step(d / d.norm(dim=1, keepdim=True) * 4.0)
        done = term | trunc
        moving = env.object[:, 3:5].norm(dim=1) > 0
        touched |= moving
        touched &= ~done; start = torch.where(done[:, None], env.object[:, :2], start)
        assert ((rew == -50) == info['wall_collision']).all() and (term == info['wall_collision']).all()
        assert ((rew == 0) == info['is_success']).all() and ((rew == -1) | (rew == 0) | (rew == -50)).all()
        assert not info['mover_collision'].any()
        # no interpenetration beyond the soft-contact allowance: centre distance >= sum of inscribed half sizes - 2 mm
        live = ~done
        assert ((env.object[live, :2] - env.pos[live]).norm(dim=1) > 0.0775 + 0.035 - 2e-3).all()
    assert touched.float().mean() > 0.3
    moved = (env.object[:, :2] - start).norm(dim=1)
    assert (moved[~touched] == 0).all() and (moved[touched] > 0).all()
    env.close()


def test_two_mover_pushing_semantics(gpr):
    """BASELINE.json config 3 (2 movers + 1 pushed object, 4 096 envs): shapes, reset rules, the mover-mover termination rule"""
    E = 4096
    env = gpr.VecBenchmarkPushingEnv(num_envs=E, num_movers=2, precision='f32')
    obs, info = env.reset(seed=2)
    assert obs['observation'].shape == (E, 8) and env.action_space.shape == (E, 4) and env.pos.shape == (E, 4)
    p = env.pos.reshape(E, 2, 2)
    assert ((p[:, 0] - p[:, 1]).norm(dim=1) > 0.22 - 1e-5).all()                      # starts apart (collision circles + offset)
    far = ((env.object[:, None, :2] - p).norm(dim=2) > env.min_mo_dist - 1e-6).all(dim=1)
    assert far.float().mean() > 0.98                                                   # PU:407 for both movers (up to max_reset_trials)
    assert torch.equal(obs['observation'][:, :4], env.pos) or (obs['observation'][:, :4] - env.pos).abs().max() < 1e-4  # positions first
    g = torch.Generator(device='cuda').manual_seed(4)
    n_mc = n_wc = n_push = 0
    for step in range(60):
        to_obj = (env.object[:, None, :2] - env.pos.reshape(E, 2, 2))
        a = (to_obj / to_obj.norm(dim=2, keepdim=True) * 5.0).reshape(E, 4)
        a = torch.where(torch.rand(E, 1, device='cuda', generator=g) < 0.3, torch.rand(E, 4, device='cuda', generator=g) * 20 - 10, a)
        obs, rew, term, trunc, info = env.step(a)
        coll = info['wall_collision'] | info['mover_collision']
        assert ((rew == -50) == coll).all() and (term == coll).all() and ((rew == 0) == info['is_success']).all()
        n_mc += int(info['mover_collision'].sum()); n_wc += int(info['wall_collision'].sum())
        n_push += int((env.object[:, 3:5].norm(dim=1) > 0).sum())
        fo = info['final_obs']['observation'][info['mover_collision']].reshape(-1, 4, 2)
        if fo.numel():
            assert ((fo[:, 0] - fo[:, 1]).norm(dim=1) <= 0.22 + 1e-3).all()           # they really met (BE:409)
    assert n_mc > 100 and n_wc > 100 and n_push > 1000, (n_mc, n_wc, n_push)
    env.close()


# ----------------------------------------------------------------------------------------------------------------------
# round 3: BenchmarkPushingEnv's own methods as written (tests/golden/pushing_methods.npz), through the C ABI
# ----------------------------------------------------------------------------------------------------------------------
def test_pushing_methods_match_reference(gpr):
    """_get_obs (PU:529-576), compute_reward / compute_terminated on arrays of info dicts (PU:457-527), reset() + _reset_callback with a
    scripted generator and the two options (PU:373-417)"""
    z = load_npz('pushing_methods.npz')
    mk = lambda n, **kw: gpr.VecBenchmarkPushingEnv(num_envs=n, precision='f64', std_noise=0.0, _push_overrides={'object_noise': 0.0}, autoreset_mode='disabled',  # noqa: E731
                                                    max_episode_steps=None, **kw)
    for mode in ('acc', 'jerk'):
        t = f'obs_{mode}'
        n = z[f'{t}_pos'].shape[0]
        env = mk(n, learn_jerk=(mode == 'jerk'), num_cycles=0)
        env.reset(seed=0)
        dev = lambda a: torch.as_tensor(a, device=env.device)  # noqa: E731
        env.pos.copy_(dev(z[f'{t}_pos'])); env.vel.copy_(dev(z[f'{t}_vel'])); env.acc.copy_(dev(z[f'{t}_acc']))
        env.object[:, :2] = dev(z[f'{t}_object']); env.goals.copy_(dev(z[f'{t}_goal']))
        obs, *_ = env.step(torch.zeros(n, 2, dtype=env.dtype, device=env.device))
        for k in ('observation', 'achieved_goal', 'desired_goal'):
            assert np.array_equal(obs[k].cpu().numpy(), z[f'{t}_{k}']), (t, k)
        env.close()
    # HER API on the reference's calling convention: an array of info dicts
    ag, dg, wc = z['reward_ag'], z['reward_dg'], z['reward_wc']
    env = mk(4, threshold_pos=float(z['reward_threshold_pos']))
    infos = np.array([{'is_success': False, 'mover_collision': False, 'wall_collision': bool(w)} for w in wc])
    assert np.array_equal(env.compute_reward(ag, dg, infos).double().cpu().numpy(), z['reward_reward'])
    assert np.array_equal(env.compute_terminated(ag, dg, infos).cpu().numpy(), z['reward_terminated'])
    r1 = env.compute_reward(ag[7], dg[7], infos[7])
    assert r1.ndim == 0 and float(r1) == z['reward_reward'][7]
    env.close()
    # reset(): scripted object candidates, with / without the options
    cand = z['reset_object_cand']
    opt_s, opt_g = z['reset_opt_start'], z['reset_opt_goal']
    for has_s in (False, True):
        for has_g in (False, True):
            idx = np.nonzero((~np.isnan(opt_s[:, 0]) == has_s) & (~np.isnan(opt_g[:, 0]) == has_g))[0]
            env = mk(len(idx))
            ms = np.where(has_s, opt_s[idx], z['reset_mover_draw'][idx]); og = np.where(has_g, opt_g[idx], z['reset_goal_draw'][idx])
            obs, info = env.reset(seed=0, options={'mover_start_xy_pos': ms.reshape(len(idx), 1, 2), 'object_goal_xy_pos': og,
                                                   'object_candidates': np.ascontiguousarray(np.swapaxes(cand[idx], 0, 1))})
            assert np.array_equal(info['candidates_used'].cpu().numpy(), z['reset_n_object'][idx]), (has_s, has_g)
            assert np.array_equal(env.pos.cpu().numpy(), z['reset_mover_start'][idx]) and np.array_equal(env.object[:, :2].cpu().numpy(), z['reset_object_start'][idx])
            assert np.array_equal(env.goals.cpu().numpy(), z['reset_object_goal'][idx])
            for k in ('observation', 'achieved_goal', 'desired_goal'):
                assert np.array_equal(obs[k].cpu().numpy(), z[f'reset_{k}'][idx]), k
            assert np.array_equal(info['wall_collision'].cpu().numpy(), z['reset_wall_collision'][idx])
            env.close()


# ----------------------------------------------------------------------------------------------------------------------
# round 4: contact model gpr-push-v3 -- known answers of rigid-body mechanics that MuJoCo obeys too (the model stays "parity unpinned":
# nothing here compares with MuJoCo's numbers, everything with what any Coulomb-friction rigid-body model must reproduce)
# ----------------------------------------------------------------------------------------------------------------------
def _push_env(gpr, E, precision, **kw):
    ov = dict(kw.pop('ov', {}))
    env = gpr.VecBenchmarkPushingEnv(num_envs=E, precision=precision, std_noise=0.0, autoreset_mode='disabled', max_episode_steps=None,
                                     _push_overrides=dict(ov, object_noise=0.0), warn_on_reset=False, **kw)
    env.reset(seed=0)
    env.vel.zero_(); env.acc.zero_(); env.mover_rot.zero_(); env.object.zero_()
    return env


@pytest.mark.parametrize('precision,rel', [('f64', 1e-9), ('f32', 2e-4)])
def test_free_sliding_object_stops_after_v2_over_2mug(gpr, precision, rel):
    """An object launched at v0 on the tile with Coulomb friction mu = 1 and no joint damping decelerates at mu g and stops after
    v0^2 / (2 mu g); the semi-implicit 1-ms integrator gives exactly v0^2 / (2 mu g) - v0 dt / 2 + (a remainder below mu g dt^2), which is
    within 2 % of the continuum answer from 0.5 m/s on. Its direction never changes and it never moves backwards."""
    E, g, mu, dt = 64, 9.81, 1.0, 1e-3
    env = _push_env(gpr, E, precision, ov={'object_damping': 0.0})
    v0 = torch.linspace(0.05, 1.2, E, dtype=torch.float64, device='cuda')
    ang = torch.linspace(0, 2 * np.pi, E, dtype=torch.float64, device='cuda')
    env.pos[:, 0] = 0.15; env.pos[:, 1] = 0.15                       # the mover parked in a corner, out of reach
    start = torch.full((E, 2), 0.40, dtype=torch.float64, device='cuda')
    env.object[:, :2] = start.to(env.dtype)
    env.object[:, 3] = (v0 * torch.cos(ang)).to(env.dtype); env.object[:, 4] = (v0 * torch.sin(ang)).to(env.dtype)
    v_in = env.object[:, 3:5].double().clone()
    sp0 = v_in.norm(dim=1)
    for _ in range(5):  # 200 ms: every object has stopped (1.2 m/s needs 122 ms)
        env.step(torch.zeros(E, 2, dtype=env.dtype, device='cuda'))
    assert float(env.object[:, 3:6].abs().max()) == 0.0
    d = (env.object[:, :2].double() - start)
    dist = d.norm(dim=1)
    n = torch.floor(sp0 / (mu * g * dt))                              # whole cycles of deceleration
    exact = dt * (n * sp0 - mu * g * dt * n * (n + 1) / 2)            # sum over cycles of dt (v0 - k mu g dt)
    assert float(((dist - exact).abs() / exact).max()) < rel
    cont = sp0 ** 2 / (2 * mu * g)
    fast = sp0 >= 0.5
    assert float(((dist - cont).abs() / cont)[fast].max()) < 0.02
    dirn = d / dist[:, None]
    assert float((dirn - v_in / sp0[:, None]).abs().max()) < (1e-9 if precision == 'f64' else 1e-4)   # straight line along v0
    record_metric(f'pushing_free_slide[{precision}]', worst_rel_error_vs_discrete=float(((dist - exact).abs() / exact).max()),
                  worst_rel_error_vs_continuum_from_0p5=float(((dist - cont).abs() / cont)[fast].max()))
    env.close()


@pytest.mark.parametrize('precision', ['f64', 'f32'])
@pytest.mark.parametrize('NM', [1, 2])
def test_steady_push_object_follows_the_mover(gpr, precision, NM):
    """A mover held at v_max by the clamp pushes the object face-on: after the impact transient the object moves with the mover's
    velocity, it turns only as far as the mover's softly held yaw gives way (K = 0.1 Nm/rad, PU:266), the mover is slowed by the friction
    force mu_tile m_o g it has to overcome only within the clamp's re-acceleration, and the interpenetration stays below a stated bound
    (20 micrometres). Faces offset laterally by up to 1.5 cm --
    still the whole object face on the mover's face -- change nothing. With two movers the second one stands aside."""
    E = 128
    env = _push_env(gpr, E, precision, v_max=0.2, num_movers=NM)
    off = torch.linspace(-0.015, 0.015, E, dtype=torch.float64, device='cuda')
    env.pos[:, 0] = 0.20; env.pos[:, 1] = 0.36
    if NM == 2:
        env.pos[:, 2] = 0.52; env.pos[:, 3] = 0.57
    env.vel[:, 0] = 0.2
    env.object[:, 0] = 0.20 + 0.0775 + 0.035 + 0.002; env.object[:, 1] = (0.36 + off).to(env.dtype)
    a = torch.zeros(E, 2 * NM, dtype=env.dtype, device='cuda'); a[:, 0] = 10.0
    worst_pen = 0.0
    for t in range(8):
        obs, rew, term, trunc, info = env.step(a)
        assert not (info['wall_collision'] | info['mover_collision']).any()
        gap = env.object[:, 0].double() - env.pos[:, 0].double() - (0.0775 + 0.035)
        if t >= 1:
            worst_pen = max(worst_pen, float((-gap).max()))
            assert float((env.object[:, 3].double() - env.vel[:, 0].double()).abs().max()) < 1e-3   # same speed as the mover
            assert float(env.object[:, 4].abs().max()) < 1e-3 and float(env.object[:, 5].abs().max()) < 5e-3 and float(env.object[:, 2].abs().max()) < 1e-3
            assert float((env.vel[:, 0].double() - 0.2).abs().max()) < 1e-3                                # the mover keeps (almost) v_max
            assert float(gap.max()) < 1e-5                                                                 # and does not lose the object
    assert worst_pen < 2e-5, worst_pen   # (centre distance: includes the corner's share of the mover's 2e-4 rad of yaw)
    assert float((env.object[:, 1].double() - (0.36 + off)).abs().max()) < 1e-4                            # no sideways drift
    record_metric(f'pushing_steady_push[{precision}-NM{NM}]', worst_penetration_m=worst_pen)
    env.close()


@pytest.mark.parametrize('precision', ['f64', 'f32'])
def test_off_centre_push_turns_the_object_the_right_way(gpr, precision):
    """The mover's face catches only the lower (upper) part of the object's face: the contact force acts below (above) the object's centre
    of mass and turns it counter-clockwise (clockwise) -- sign(omega) = sign of the lateral offset of the object against the mover -- at a
    rate bounded by v / (object half size); the object is still pushed ahead. Symmetric offsets give mirror-image motions."""
    E = 64
    env = _push_env(gpr, E, precision, v_max=0.2)
    mag = torch.linspace(0.085, 0.105, E // 2, dtype=torch.float64, device='cuda')   # object centre beyond the mover's corner (0.0775): partial face contact
    off = torch.cat([mag, -mag])
    env.pos[:, 0] = 0.20; env.pos[:, 1] = 0.36
    env.vel[:, 0] = 0.2
    env.object[:, 0] = 0.20 + 0.0775 + 0.035 + 0.002; env.object[:, 1] = (0.36 + off).to(env.dtype)
    a = torch.zeros(E, 2, dtype=env.dtype, device='cuda'); a[:, 0] = 10.0
    for t in range(6):
        env.step(a)
        w = env.object[:, 5].double()
        if t >= 1:
            assert (torch.sign(w) == torch.sign(off)).all(), t
            assert float(w.abs().max()) < 0.2 / 0.035 and float(w.abs().min()) > 0.01
            assert float(env.object[:, 3].min()) > 0.1                                                     # pushed ahead
    th, y = env.object[:, 2].double(), env.object[:, 1].double() - 0.36
    tol = 1e-9 if precision == 'f64' else 2e-3
    assert float((th[:E // 2] + th[E // 2:]).abs().max()) < tol and float((y[:E // 2] + y[E // 2:]).abs().max()) < tol  # mirror symmetry
    env.close()


@pytest.mark.parametrize('precision', ['f64', 'f32'])
def test_contact_resists_acceleration_with_bounded_penetration(gpr, precision):
    """A mover accelerating at a_max into the resting object: the object is taken along (no tunnelling), and the steady interpenetration
    stays below a_max dt tc = 0.2 mm: every cycle the acceleration rebuilds an approach speed of a dt, the solve removes it down to the
    recovery bias depth / tc, so the depth settles where depth / tc balances it."""
    E = 32
    env = _push_env(gpr, E, precision, v_max=2.0)
    env.pos[:, 0] = 0.15; env.pos[:, 1] = 0.36
    env.object[:, 0] = 0.15 + 0.0775 + 0.035 + torch.linspace(1e-4, 3e-3, E, device='cuda').to(env.dtype); env.object[:, 1] = 0.36
    a = torch.zeros(E, 2, dtype=env.dtype, device='cuda'); a[:, 0] = 10.0
    worst = 0.0
    for t in range(5):
        obs, rew, term, trunc, info = env.step(a)
        gap = env.object[:, 0].double() - env.pos[:, 0].double() - (0.0775 + 0.035)
        worst = max(worst, float((-gap).max()))
        if t >= 2:
            assert float((env.object[:, 3].double() - env.vel[:, 0].double()).abs().max()) < 5e-3
    assert worst < 2e-4, worst
    assert float(env.object[:, 0].min()) > 0.15 + 0.0775 + 0.035 + 0.05    # carried along for 5 steps of acceleration
    record_metric(f'pushing_accelerated_push[{precision}]', worst_penetration_m=worst)
    env.close()
