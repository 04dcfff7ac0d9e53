"""Genetic algorithm whose unit of work is a GENERATION: breed an offspring matrix, score it with one device pass.

Counterpart of the reference's ``genetic_algorithm.Optimiser`` with its default operator pipeline
(src/IMCoalHMM/genetic_algorithm.py: uniform start :186-210, tournament selection :342-366, one-point crossover
:426-444, Gaussian point mutation :620-640, elitism + hall of fame :715-730, the generation loop :782-836), built for a
likelihood that is evaluated in batches (``Likelihood.batch``: all offspring of a generation share one pass over the
alignment chunks on the GPU).

Design (not the reference's object-per-individual loop):

* a population is three arrays - ``genomes (P, D)``, ``fitness (P,)``, ``serial (P,)`` (a birth number: identity);
* ``breed`` turns the current arrays into the next generation's offspring matrix ``(P - elite, D)`` WITHOUT any fitness
  call - selection and elitism only look at the parents' fitness - and ``maximise`` scores that matrix with one
  ``fitness_batch`` call;
* the hall of fame is a pair of arrays kept sorted by insertion.

Random numbers are drawn from a ``random.Random`` in exactly the order the reference draws them (tournament starts,
then per offspring the crossover point followed by one uniform - and, if it hits, one gauss - per allele), and ties are
broken as a stable descending sort breaks them, so a trajectory recorded from the reference's own class
(tests/golden/ga_golden.json) is reproduced bit for bit (tests/test_ga_cpu.py).  The reference is Python 2, whose
``round`` goes half away from zero: ``_round_half_up`` keeps that for the two size formulas.
"""
import collections
import datetime
import math
import random as _random

import numpy as np

Member = collections.namedtuple("Member", "genome fitness")     # what iterating a population / the hall of fame yields


class ExitCondition(object):
    ABORT = 'ABORT'
    GENERATIONS = 'GENERATIONS'
    TIMEOUT = 'TIMEOUT'


def _round_half_up(x):
    return int(math.floor(x + 0.5))


def _descending(fitness):
    """Indices by decreasing fitness; equal values keep their order (what ``sort(key=fitness, reverse=True)`` gives)."""
    return np.argsort(-np.asarray(fitness, dtype=np.float64), kind="stable")


class Scored(object):
    """Genomes with their fitness and birth numbers, as arrays; iterates as ``Member(genome tuple, fitness float)``."""

    def __init__(self, genomes, fitness, serial):
        self.genomes = np.asarray(genomes, dtype=np.float64)            # (P, D)
        self.fitness = np.asarray(fitness, dtype=np.float64)
        self.serial = np.asarray(serial, dtype=np.int64)

    def __len__(self):
        return self.fitness.shape[0]

    def __getitem__(self, k):
        return Member(tuple(self.genomes[k].tolist()), float(self.fitness[k]))

    def __iter__(self):
        return (self[k] for k in range(len(self)))

    def take(self, index):
        return Scored(self.genomes[index], self.fitness[index], self.serial[index])

    @staticmethod
    def join(a, b):
        return Scored(np.concatenate([a.genomes, b.genomes]), np.concatenate([a.fitness, b.fitness]),
                      np.concatenate([a.serial, b.serial]))


class HallOfFame(Scored):
    """The best ``capacity`` members ever seen, best first.  A member enters once (by birth number); among equal
    fitness the earlier entrant stays in front (genetic_algorithm.py:715-730)."""

    def __init__(self, capacity, genome_length):
        Scored.__init__(self, np.empty((0, genome_length)), np.empty(0), np.empty(0, dtype=np.int64))
        self.capacity = capacity

    def offer(self, scored):
        for k in range(len(scored)):
            if scored.serial[k] in self.serial:
                continue
            at = int(np.searchsorted(-self.fitness, -scored.fitness[k], side="right"))   # behind its equals
            if at >= self.capacity:
                continue
            self.genomes = np.insert(self.genomes, at, scored.genomes[k], axis=0)[:self.capacity]
            self.fitness = np.insert(self.fitness, at, scored.fitness[k])[:self.capacity]
            self.serial = np.insert(self.serial, at, scored.serial[k])[:self.capacity]


class Context(object):
    """State handed to ``log`` every generation and returned by ``maximise``: ``generation``, ``population``,
    ``hall_of_fame``, ``evaluations``, ``start`` / ``elapsed``, ``exit_condition``; ``log`` may set ``aborted``."""

    def __init__(self, optimiser, genome_length):
        self.optimiser, self.start, self.elapsed = optimiser, datetime.datetime.now(), datetime.timedelta(0)
        self.generation = self.evaluations = 0
        self.aborted, self.exit_condition = False, None             # aborted -> the run ends with ExitCondition.ABORT
        self.population = Scored(np.empty((0, genome_length)), [], [])
        self.hall_of_fame = HallOfFame(optimiser.hall_of_fame_size, genome_length)

    def tick(self):
        self.elapsed = datetime.datetime.now() - self.start
        self.generation += 1


class Optimiser(object):
    """``Optimiser(rng=random)`` with the reference's tunables and defaults (:745-755, :228, :339, :557, :617-618)."""

    TUNABLES = dict(
        population_size=100, elite_count=1, hall_of_fame_size=5,
        max_generations=500, timeout=None,         # limits: generations, a datetime.timedelta (None: no limit)
        log=None,                                  # callable(context), once per generation
        selection_ratio=0.75,                      # breeders per generation, as a share of the population
        tournament_ratio=0.1,                      # tournament window, as a share of the population
        point_mutation_ratio=0.15,                 # chance that an allele is perturbed ...
        mutate=True,                               # ... False: no mutation and no draws (the reference's ``mutation = None``)
        mutation_mu=0.0, mutation_sigma=0.01)

    def __init__(self, rng=_random):
        self.random = rng
        self._births = 0
        for name, default in self.TUNABLES.items():
            setattr(self, name, default)

    # ---- operators on arrays ------------------------------------------------------------------------------------
    def initial_genomes(self, genome_length):
        """(P, D) uniform in [0, 1), drawn row by row."""
        draw = self.random.uniform
        return np.array([[draw(0.0, 1.0) for _ in range(genome_length)] for _ in range(self.population_size)])

    def tournament_winners(self, fitness):
        """Indices of the breeders: per breeder one window start is drawn; the window's first best member wins."""
        size = len(fitness)
        wanted = max(1, _round_half_up(size * self.selection_ratio))
        window = _round_half_up(size * self.tournament_ratio)
        starts = [self.random.randint(0, size - window) for _ in range(wanted)]
        starts = [min(a, size - 1) for a in starts]                    # (a zero-width window may start one past the end)
        return np.array([a + int(np.argmax(fitness[a:a + max(window, 1)])) for a in starts], dtype=np.int64)

    def perturb(self, row):
        """In-place Gaussian point mutation of one genome, clamped to [0, 1]; draws per allele, left to right."""
        rnd = self.random
        for d in range(row.shape[0]):
            if rnd.uniform(0.0, 1.0) < self.point_mutation_ratio:
                row[d] = min(max(0.0, row[d] + rnd.gauss(self.mutation_mu, self.mutation_sigma)), 1.0)

    def breed(self, parents):
        """(elite indices, offspring matrix) for the generation after ``parents`` - no fitness is evaluated here."""
        fit = parents.fitness
        winners = self.tournament_winners(fit)
        mates = winners[_descending(fit[winners])]                   # breeders, best first
        elite = _descending(fit)[:self.elite_count]
        n_children = self.population_size - len(elite)
        n_mates, depth = len(mates), parents.genomes.shape[1]
        children = np.empty((n_children, depth))
        for c in range(n_children):
            # two breeders half a ring apart; the fitter one (on a tie the first) gives the head of the genome
            x, y = mates[c % n_mates], mates[(c + n_mates // 2) % n_mates]
            head, tail = (y, x) if fit[y] > fit[x] else (x, y)
            cut = self.random.randint(1, depth - 1)
            children[c, :cut] = parents.genomes[head, :cut]
            children[c, cut:] = parents.genomes[tail, cut:]
            if self.mutate:
                self.perturb(children[c])
        return elite, children

    # ---- the loop -----------------------------------------------------------------------------------------------
    def _score(self, context, fitness_batch, genomes):
        values = np.asarray(list(fitness_batch([row.tolist() for row in genomes])), dtype=np.float64)
        if values.shape != (len(genomes),):
            raise ValueError("fitness_batch must return one value per genome")
        values[np.isnan(values)] = -np.inf                           # NaN = "very bad" (:29-30)
        serial = np.arange(self._births, self._births + len(genomes))
        self._births += len(genomes)
        context.evaluations += len(genomes)
        return Scored(genomes, values, serial)

    def _verdict(self, context):
        over = lambda value, limit: limit is not None and value > limit          # noqa: E731
        checks = ((context.aborted, ExitCondition.ABORT),
                  (over(context.generation + 1, self.max_generations), ExitCondition.GENERATIONS),
                  (over(context.elapsed, self.timeout), ExitCondition.TIMEOUT))
        return next((verdict for hit, verdict in checks if hit), None)

    def maximise(self, fitness_batch, genome_length):
        """``fitness_batch(list of genomes) -> one float per genome``; returns the final ``Context``."""
        if not callable(fitness_batch):
            raise ValueError("fitness_batch must be callable")
        if genome_length <= 0 or not 0 <= self.elite_count < self.population_size:
            raise ValueError("need a positive genome length and 0 <= elite_count < population_size")
        context = Context(self, genome_length)
        context.population = self._score(context, fitness_batch, self.initial_genomes(genome_length))
        while context.exit_condition is None:
            context.hall_of_fame.offer(context.population)
            context.tick()
            if callable(self.log):
                self.log(context)
            context.exit_condition = self._verdict(context)
            if context.exit_condition is None:
                elite, children = self.breed(context.population)
                context.population = Scored.join(context.population.take(elite), self._score(context, fitness_batch, children))
        return context
