"""imcoalhmm_amd.ga against a trajectory recorded from the REFERENCE's own genetic algorithm
(tests/golden/ga_golden.json, generator tests/golden/make_ga_fixture.py): breeding a whole generation before
evaluating it in one batch consumes the random stream in the reference's order, so populations, best genomes and the
hall of fame must match bit for bit."""
import json
import math
import os
import random

import numpy as np

from imcoalhmm_amd import ga

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ga_golden.json")


def fitness(genome):
    return -sum((x - 0.3 - 0.1 * k) ** 2 for k, x in enumerate(genome)) + 0.05 * math.cos(25.0 * genome[0])


def _run(gold, rng):
    calls = []

    def fitness_batch(genomes):
        calls.append(len(genomes))
        return [fitness(g) for g in genomes]

    opt = ga.Optimiser(rng)
    opt.population_size = gold["population_size"]
    opt.max_generations = gold["max_generations"]
    opt.elite_count = gold["elite_count"]
    opt.hall_of_fame_size = gold["hall_of_fame_size"]
    seen = []
    opt.log = lambda c: seen.append({
        "generation": c.generation, "fitness": [i.fitness for i in c.population],
        "best_genome": list(max(c.population, key=lambda i: i.fitness).genome),
        "hall_of_fame": [[i.fitness, list(i.genome)] for i in c.hall_of_fame]})
    ctx = opt.maximise(fitness_batch, gold["genome_length"])
    return ctx, seen, calls


def test_reproduces_the_reference_trajectory():
    gold = json.load(open(GOLDEN))
    ctx, seen, calls = _run(gold, random.Random(gold["seed"]))
    assert ctx.exit_condition == gold["exit_condition"] == ga.ExitCondition.GENERATIONS
    assert len(seen) == len(gold["generations"])
    for got, want in zip(seen, gold["generations"]):
        assert got == want                                   # floats compared exactly
    # one batch for the initial population, then one per generation of (population - elite) offspring
    assert calls == [gold["population_size"]] + [gold["population_size"] - gold["elite_count"]] * (gold["max_generations"] - 1)
    assert ctx.evaluations == sum(calls)


def test_nan_fitness_is_minus_infinity_and_abort():
    opt = ga.Optimiser(random.Random(3))
    opt.population_size = 10
    opt.max_generations = 50

    def batch(genomes):
        return [float("nan") if g[0] > 0.5 else -abs(g[0] - 0.25) for g in genomes]

    def log(c):
        if c.generation == 4:
            c.aborted = True
    opt.log = log
    ctx = opt.maximise(batch, 2)
    assert ctx.exit_condition == ga.ExitCondition.ABORT and ctx.generation == 4
    assert all(not math.isnan(i.fitness) for i in ctx.population)
    assert ctx.hall_of_fame[0].fitness == max(i.fitness for i in ctx.hall_of_fame)


def test_batch_size_mismatch_raises():
    import pytest
    opt = ga.Optimiser(random.Random(1))
    opt.population_size = 4
    with pytest.raises(ValueError):
        opt.maximise(lambda genomes: [0.0], 3)
    with pytest.raises(ValueError):
        opt.maximise(None, 3)
    assert np.isfinite(fitness([0.1, 0.2]))
