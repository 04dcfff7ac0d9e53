"""Genetic-algorithm driver with one batched fitness call per generation.

Counterpart of the reference's ``genetic_algorithm.Optimiser.maximise`` (src/IMCoalHMM/genetic_algorithm.py:757-839)
with its default pipeline - uniform initialisation (:186-210), tournament selection (:342-366), one-point crossover
(:426-444), Gaussian mutation (:620-640), elitism and a hall of fame (:715-730) - and the same tunables.

The reference evaluates every offspring as soon as it is bred (:829).  The breeding of a generation never looks at the
fitness of that generation's own offspring (breeders and elite come from the previous population, :801-813), and the
fitness function draws no random numbers, so generating all genomes of a generation first and evaluating them in ONE
``fitness_batch(genomes) -> fitnesses`` call (e.g. ``Likelihood.batch`` behind the parameter transform of
scripts/heuristic-optimiser.py:229-231) consumes the random stream in exactly the reference's order and produces exactly
the reference's populations: tests/golden/ga_golden.json was recorded from the reference's own class and is reproduced
bit for bit (tests/test_ga_cpu.py).
"""
import datetime
import math
import random as _random


class ExitCondition(object):
    ABORT = 'ABORT'
    GENERATIONS = 'GENERATIONS'
    TIMEOUT = 'TIMEOUT'


class Individual(object):
    """A genome (tuple) with its fitness; NaN fitness counts as -inf (genetic_algorithm.py:29-30)."""
    __slots__ = ("genome", "fitness")

    def __init__(self, genome, fitness):
        self.genome = tuple(genome)
        fitness = float(fitness)
        self.fitness = float('-inf') if math.isnan(fitness) else fitness

    def __str__(self):
        return '{0}:[{1}]'.format(self.fitness, ' '.join(map(str, self.genome)))


class UniformInitialisation(object):
    def __init__(self, rng=_random):
        self.random = rng

    def genomes(self, population_size, genome_length):
        return [[self.random.uniform(0.0, 1.0) for _ in range(genome_length)] for _ in range(population_size)]


class TournamentSelection(object):
    def __init__(self, rng=_random):
        self.random = rng
        self.selection_ratio = 0.75
        self.tournament_ratio = 0.1

    def select(self, population):
        size = max(1, int(round(float(len(population)) * self.selection_ratio)))
        tournament_size = int(round(len(population) * self.tournament_ratio))
        breeders = []
        while len(breeders) < size:
            a = self.random.randint(0, len(population) - tournament_size)
            winner = population[a]
            for individual in population[a + 1:a + tournament_size]:
                if individual.fitness > winner.fitness:
                    winner = individual
            breeders.append(winner)
        return breeders


class OnePointCrossover(object):
    def __init__(self, rng=_random):
        self.random = rng

    def crossover(self, individuals):
        left, right = individuals
        i = self.random.randint(1, len(left.genome) - 1)
        return list(left.genome[:i] + right.genome[i:])


class GaussianMutation(object):
    def __init__(self, rng=_random):
        self.random = rng
        self.point_mutation_ratio = 0.15
        self.mu = 0.0
        self.sigma = 0.01

    def mutate(self, genome0):
        genome = []
        for allele in genome0:
            if self.random.uniform(0.0, 1.0) >= self.point_mutation_ratio:
                genome.append(allele)
                continue
            genome.append(min(max(0.0, allele + self.random.gauss(self.mu, self.sigma)), 1.0))
        return genome


class Context(object):
    def __init__(self, optimiser):
        self.aborted = False
        self.elapsed = datetime.timedelta(seconds=0)
        self.exit_condition = None
        self.generation = 0
        self.hall_of_fame = []
        self.optimiser = optimiser
        self.population = []
        self.start = datetime.datetime.now()
        self.evaluations = 0

    def submit_to_hall_of_fame(self, individual, max_size):
        if individual in self.hall_of_fame:
            return
        self.hall_of_fame.append(individual)
        list.sort(self.hall_of_fame, key=lambda x: x.fitness, reverse=True)
        self.hall_of_fame = self.hall_of_fame[:max_size]


class Optimiser(object):
    """``Optimiser(rng=random)``; operators may be replaced by any objects with the same methods."""

    def __init__(self, rng=_random):
        self.mutation = GaussianMutation(rng)
        self.crossover = OnePointCrossover(rng)
        self.initialisation = UniformInitialisation(rng)
        self.selection = TournamentSelection(rng)
        self.population_size = 100
        self.hall_of_fame_size = 5
        self.elite_count = 1
        self.log = None
        self.max_generations = 500
        self.timeout = None

    @staticmethod
    def _evaluate(context, fitness_batch, genomes):
        values = list(fitness_batch(genomes))
        if len(values) != len(genomes):
            raise ValueError("fitness_batch must return one value per genome")
        context.evaluations += len(genomes)
        return [Individual(g, f) for g, f in zip(genomes, values)]

    def maximise(self, fitness_batch, genome_length):
        """``fitness_batch(list of genomes) -> sequence of floats``; returns the final ``Context``."""
        if not callable(fitness_batch) or genome_length <= 0 or not self.elite_count < self.population_size:
            raise ValueError("need a callable fitness, a positive genome length and elite_count < population_size")
        context = Context(self)
        context.population = self._evaluate(context, fitness_batch,
                                            self.initialisation.genomes(self.population_size, genome_length))
        for individual in context.population:
            context.submit_to_hall_of_fame(individual, self.hall_of_fame_size)
        while True:
            context.elapsed = datetime.datetime.now() - context.start
            context.generation += 1
            if self.log is not None:
                self.log(context)
            if context.aborted:
                context.exit_condition = ExitCondition.ABORT
                return context
            if self.max_generations is not None and context.generation >= self.max_generations:
                context.exit_condition = ExitCondition.GENERATIONS
                return context
            if self.timeout is not None and context.elapsed > self.timeout:
                context.exit_condition = ExitCondition.TIMEOUT
                return context
            breeders = self.selection.select(context.population)
            list.sort(breeders, key=lambda x: x.fitness, reverse=True)
            elite = sorted(context.population, key=lambda x: x.fitness, reverse=True)[:self.elite_count]
            # breed the whole generation, then evaluate it in one batch
            i, j = 0, len(breeders) // 2
            genomes = []
            while len(elite) + len(genomes) < self.population_size:
                genome = self.crossover.crossover(sorted((breeders[i], breeders[j]), key=lambda x: x.fitness, reverse=True))
                if self.mutation is not None:
                    genome = self.mutation.mutate(genome)
                genomes.append(genome)
                i = (i + 1) % len(breeders)
                j = (j + 1) % len(breeders)
            context.population = elite + self._evaluate(context, fitness_batch, genomes)
            for individual in context.population:
                context.submit_to_hall_of_fame(individual, self.hall_of_fame_size)
