"""Particle swarm optimisation with one batched fitness call per iteration.

Counterpart of src/IMCoalHMM/particle_swarm.py: the same update rule and the same tunables (``omega``,
``phi_particle``, ``phi_swarm``, ``particle_count``, ``max_iterations``, ``max_initial_velocity``, ``timeout``,
``log``), positions initialised uniformly in [0, 1) per dimension and NaN fitness mapped to -inf
(particle_swarm.py:113-117,122-131,168-192) - but the swarm is held as arrays and the whole population is
evaluated by ONE call ``batch_fitness(positions[P, D]) -> fitness[P]`` (e.g. a ``Likelihood.batch`` behind a
parameter transform), which is what keeps a GPU busy.

Documented difference: the reference updates the swarm's best inside the particle loop, so particle k of an
iteration already sees improvements found by particles < k of the same iteration (particle_swarm.py:186-192);
here the swarm best moves once per iteration, after the batch.  Both are standard PSO variants.
"""
import datetime

import numpy as np


class ExitCondition(object):
    ABORT = 'ABORT'
    ITERATIONS = 'ITERATIONS'
    TIMEOUT = 'TIMEOUT'


class Context(object):
    """State handed to ``log`` every iteration and returned by ``maximise``."""

    def __init__(self, optimiser):
        self.optimiser = optimiser
        self.aborted = False            # a log callback may set this to stop the run
        self.iteration = 0
        self.start = datetime.datetime.now()
        self.elapsed = datetime.timedelta(seconds=0)
        self.exit_condition = None
        self.positions = None           # [P, D] current
        self.velocities = None          # [P, D]
        self.fitness = None             # [P] current
        self.best_positions = None      # [P, D] per particle
        self.best_fitness = None        # [P]
        self.swarm_best_position = None
        self.swarm_best_fitness = None
        self.evaluations = 0


class Optimiser(object):
    def __init__(self, seed=None):
        self.omega = 0.9
        self.phi_particle = 0.3
        self.phi_swarm = 0.1
        self.log = None
        self.max_iterations = 500
        self.max_initial_velocity = 0.02
        self.particle_count = 100
        self.timeout = None
        self.rng = np.random.default_rng(seed)

    def _evaluate(self, context, batch_fitness, positions):
        values = np.asarray(batch_fitness(positions), dtype=np.float64).reshape(-1)
        if values.shape[0] != positions.shape[0]:
            raise ValueError("batch_fitness must return one value per particle")
        context.evaluations += positions.shape[0]
        return np.where(np.isnan(values), -np.inf, values)

    def maximise(self, batch_fitness, parameter_count):
        """Run the swarm; returns the final ``Context``."""
        if not callable(batch_fitness) or parameter_count <= 0:
            raise ValueError("need a callable fitness and a positive parameter count")
        P, D = int(self.particle_count), int(parameter_count)
        ctx = Context(self)
        ctx.positions = self.rng.uniform(0.0, 1.0, size=(P, D))
        ctx.velocities = self.rng.uniform(-self.max_initial_velocity, self.max_initial_velocity, size=(P, D))
        ctx.fitness = self._evaluate(ctx, batch_fitness, ctx.positions)
        ctx.best_positions = ctx.positions.copy()
        ctx.best_fitness = ctx.fitness.copy()
        top = int(np.argmax(ctx.best_fitness))
        ctx.swarm_best_position = ctx.best_positions[top].copy()
        ctx.swarm_best_fitness = float(ctx.best_fitness[top])
        while True:
            ctx.elapsed = datetime.datetime.now() - ctx.start
            ctx.iteration += 1
            if self.log is not None:
                self.log(ctx)
            if ctx.aborted:
                ctx.exit_condition = ExitCondition.ABORT
                return ctx
            if self.max_iterations is not None and ctx.iteration >= self.max_iterations:
                ctx.exit_condition = ExitCondition.ITERATIONS
                return ctx
            if self.timeout is not None and ctx.elapsed > self.timeout:
                ctx.exit_condition = ExitCondition.TIMEOUT
                return ctx
            r_particle = self.rng.uniform(0.0, 1.0, size=(P, 1))     # one draw per particle, shared by its dimensions
            r_swarm = self.rng.uniform(0.0, 1.0, size=(P, 1))
            ctx.velocities = (self.omega * ctx.velocities
                              + self.phi_particle * r_particle * (ctx.best_positions - ctx.positions)
                              + self.phi_swarm * r_swarm * (ctx.swarm_best_position[None, :] - ctx.positions))
            ctx.positions = ctx.positions + ctx.velocities
            ctx.fitness = self._evaluate(ctx, batch_fitness, ctx.positions)
            better = ctx.fitness > ctx.best_fitness
            ctx.best_positions[better] = ctx.positions[better]
            ctx.best_fitness[better] = ctx.fitness[better]
            top = int(np.argmax(ctx.best_fitness))
            if ctx.best_fitness[top] > ctx.swarm_best_fitness:
                ctx.swarm_best_fitness = float(ctx.best_fitness[top])
                ctx.swarm_best_position = ctx.best_positions[top].copy()
