#!/usr/bin/env python3
"""Generate tests/golden/ga_golden.json: a trajectory of the REFERENCE's genetic algorithm.

Pins imcoalhmm_amd/ga.py.  As in make_fixtures.py, a scratch copy of the reference's
Python-2 package under /tmp is converted with the stock ``lib2to3`` tool and ``IMCoalHMM.genetic_algorithm`` is imported
from there (never committed, never shipped).  The fitness is a closed-form test function, the random stream is the
``random`` module seeded below; what is written is data only: per generation the population's fitness values, the best
genome and the hall of fame.

Run once in the build container:  python tests/golden/make_ga_fixture.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_fixtures import import_reference_models  # noqa: E402


def fitness(genome):
    """Smooth, multimodal, deterministic; also used (vectorised over genomes) by the test."""
    import math
    return -sum((x - 0.3 - 0.1 * k) ** 2 for k, x in enumerate(genome)) + 0.05 * math.cos(25.0 * genome[0])


def main():
    import_reference_models()                     # sets up the converted scratch copy on sys.path
    from IMCoalHMM import genetic_algorithm as ref
    out = {"seed": 20240901, "population_size": 24, "genome_length": 5, "max_generations": 9, "elite_count": 2,
           "hall_of_fame_size": 4, "generations": []}
    random.seed(out["seed"])
    opt = ref.Optimiser()
    opt.population_size = out["population_size"]
    opt.max_generations = out["max_generations"]
    opt.elite_count = out["elite_count"]
    opt.hall_of_fame_size = out["hall_of_fame_size"]

    def log(context):
        best = max(context.population, key=lambda ind: ind.fitness)
        out["generations"].append({
            "generation": context.generation,
            "fitness": [ind.fitness for ind in context.population],
            "best_genome": list(best.genome),
            "hall_of_fame": [[ind.fitness, list(ind.genome)] for ind in context.hall_of_fame]})
    opt.log = log
    ctx = opt.maximise(fitness, out["genome_length"])
    out["exit_condition"] = ctx.exit_condition
    with open(os.path.join(HERE, "ga_golden.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote ga_golden.json: %d generations, best %.12g" % (len(out["generations"]), ctx.hall_of_fame[0].fitness))


if __name__ == "__main__":
    main()
