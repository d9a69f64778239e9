"""ORACLE (test infrastructure): the thermophysical package the reference selects for its gas phase,
    hePsiThermo<reactingMixture/singleStepReactingMixture<sutherland<janaf<perfectGas<specie>>>>, sensibleEnthalpy>
(cases/steckler/constant/thermophysicalProperties:18-27; coefficients cases/steckler/constant/thermo.compressibleGas,
reaction cases/steckler/constant/reactions), restated in numpy from OpenFOAM-dev @940e28f (not vendored in /root/reference):
  src/thermophysicalModels/specie/thermo/janaf/janafThermoI.H          Cp, Ha, Hs, Hc, coefficient mixing (operator+=)
  src/thermophysicalModels/specie/equationOfState/perfectGas/perfectGasI.H   rho, psi, CpMCv
  src/thermophysicalModels/specie/transport/sutherland/sutherlandTransportI.H   mu, kappa (modified Eucken), alphah
  src/thermophysicalModels/specie/thermo/thermo/thermoI.H              T(h) Newton iteration (tol 1e-4, start at the old T)
  src/thermophysicalModels/specie/specie/specieI.H                     mass-fraction weighted molecular weight
  src/thermophysicalModels/reactionThermo/mixtures/multiComponentMixture/multiComponentMixture.C   cellMixture: progressive sum
  src/thermophysicalModels/reactionThermo/mixtures/singleStepReactingMixture/singleStepReactingMixture.C   qFuel, s, stoicRatio, Yprod0
  src/thermophysicalModels/basic/psiThermo/hePsiThermo.C               calculate(): T, psi, mu, alpha cell by cell / face by face
Pin: the numbers the reference prints at start-up (cases/steckler/original/linux64/log.fireFoam:46-52,108):
  Fuel heat of combustion :46357151, stoichiometric air-fuel ratio :15.571544, stoichiometric oxygen-fuel ratio :3.6282945,
  maximum products H2O 0.098613587 / CO2 0.18067909 / N2 0.72070733, stoichiometric mixture fraction 0.060344407
-- tests/test_thermo_cpu.py.  Thermo is mass based at this OpenFOAM commit (coefficients are multiplied by R = RR/W).
Only tests/ may import this module."""
import re

import numpy as np

RR = 6.0221417930e26 * 1.38065e-23      # J/(kmol K): NA*k of OpenFOAM-dev etc/controlDict (SI)
TSTD = 298.15
SMALL = 1.0e-15


def parse_thermo_file(path):
    """species dictionaries of a foamChemistryThermoFile: {name: dict(W, Tlow, Thigh, Tcommon, high[7], low[7], As, Ts)}"""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    out = {}
    i = txt.index("}") + 1 if "FoamFile" in txt else 0                     # skip the header
    for mt in re.finditer(r"\n(\w+)\s*\{", txt[i:]):
        name = mt.group(1)
        if name in ("specie", "thermodynamics", "transport", "elements"):
            continue
        start = i + mt.end()
        depth, j = 1, start
        while depth:
            ch = txt[j]
            depth += ch == "{"; depth -= ch == "}"; j += 1
        body = txt[start:j]
        num = lambda key: float(re.search(r"\b%s\s+([-+0-9.eE]+)\s*;" % key, body).group(1))
        vec = lambda key: np.array([float(x) for x in re.search(r"\b%s\s*\(([^)]*)\)" % key, body).group(1).split()])
        out[name] = dict(W=num("molWeight"), Tlow=num("Tlow"), Thigh=num("Thigh"), Tcommon=num("Tcommon"), high=vec("highCpCoeffs"),
                         low=vec("lowCpCoeffs"), As=num("As"), Ts=num("Ts"))
    return out


class Mixture:
    """arrays of sutherland<janaf<perfectGas<specie>>> objects: one per cell / face (the result of cellMixture / patchFaceMixture)"""

    def __init__(self, W, Tlow, Thigh, Tcommon, high, low, As, Ts):
        self.W, self.Tlow, self.Thigh, self.Tcommon, self.high, self.low, self.As, self.Ts = W, Tlow, Thigh, Tcommon, high, low, As, Ts

    # ---- perfectGas
    def R(self):
        return RR / self.W

    def psi(self, p, T):
        return 1.0 / (self.R() * T)

    def rho(self, p, T):
        return p / (self.R() * T)

    # ---- janaf
    def _coeffs(self, T):
        return np.where((np.asarray(T) < self.Tcommon)[..., None], self.low, self.high)

    def limit(self, T):
        return np.minimum(np.maximum(T, self.Tlow), self.Thigh)

    def Cp(self, p, T):
        a = self._coeffs(T)
        return ((((a[..., 4] * T + a[..., 3]) * T + a[..., 2]) * T + a[..., 1]) * T + a[..., 0])

    def Ha(self, p, T):
        a = self._coeffs(T)
        return ((((a[..., 4] / 5.0 * T + a[..., 3] / 4.0) * T + a[..., 2] / 3.0) * T + a[..., 1] / 2.0) * T + a[..., 0]) * T + a[..., 5]

    def Hc(self):
        a = self.low
        return ((((a[..., 4] / 5.0 * TSTD + a[..., 3] / 4.0) * TSTD + a[..., 2] / 3.0) * TSTD + a[..., 1] / 2.0) * TSTD + a[..., 0]) * TSTD + a[..., 5]

    def Hs(self, p, T):
        return self.Ha(p, T) - self.Hc()

    def Cv(self, p, T):
        return self.Cp(p, T) - self.R()                  # perfectGas::CpMCv = R

    # ---- thermo<...>::THs: Newton iteration from T0, |dT| <= T0*1e-4
    def THs(self, h, p, T0, tol=1.0e-4, maxIter=100):
        h = np.asarray(h, float); T0 = np.asarray(T0, float) * np.ones_like(h)
        Tnew = T0.copy()
        Ttol = T0 * tol
        active = np.ones(h.shape, bool)
        for _ in range(maxIter + 1):
            Test = Tnew.copy()
            step = self.limit(Test - (self.Hs(p, Test) - h) / self.Cp(p, Test))
            Tnew = np.where(active, step, Tnew)
            active = active & (np.abs(Tnew - Test) > Ttol)
            if not active.any():
                return Tnew
        raise RuntimeError("THs: maximum number of iterations exceeded")

    # ---- sutherland
    def mu(self, p, T):
        return self.As * np.sqrt(T) / (1.0 + self.Ts / T)

    def kappa(self, p, T):
        Cv = self.Cv(p, T)
        return self.mu(p, T) * Cv * (1.32 + 1.77 * self.R() / Cv)

    def alphah(self, p, T):
        return self.kappa(p, T) / self.Cp(p, T)


class Species:
    """speciesData of a multiComponentMixture, in the order of the `species (...)` list of the chemistry file"""

    def __init__(self, names, table):
        self.names = list(names)
        self.d = [table[n] for n in names]
        self.W = np.array([s["W"] for s in self.d])
        # janafThermo constructor: coefficients *= R (mass based)
        self.high = np.stack([s["high"] * (RR / s["W"]) for s in self.d])
        self.low = np.stack([s["low"] * (RR / s["W"]) for s in self.d])

    def single(self, i):
        s = self.d[i]
        return Mixture(s["W"], s["Tlow"], s["Thigh"], s["Tcommon"], self.high[i], self.low[i], s["As"], s["Ts"])

    def mixture(self, Y):
        """cellMixture: mixture = Y0*specie0; mixture += Yn*specie_n ...  Y[nSpecies][n] -> one Mixture holding n objects"""
        Y = np.asarray(Y, float)
        if Y.ndim == 1:
            Y = Y[:, None]
        n = Y.shape[1]
        d0 = self.d[0]
        sumY = Y[0].copy()
        W = np.full(n, d0["W"])
        Tlow = np.full(n, d0["Tlow"]); Thigh = np.full(n, d0["Thigh"]); Tcommon = np.full(n, d0["Tcommon"])
        high = np.tile(self.high[0], (n, 1)); low = np.tile(self.low[0], (n, 1))
        As = np.full(n, d0["As"]); Ts = np.full(n, d0["Ts"])
        for k in range(1, len(self.d)):
            s, Yk = self.d[k], Y[k]
            # sutherlandTransport::operator+= (its Y1 is taken before the base classes add)
            Y1 = sumY.copy()
            new = sumY + Yk
            ok = np.abs(new) > SMALL
            safe = np.where(ok, new, 1.0)
            # specie::operator+=
            with np.errstate(divide="ignore", invalid="ignore"):            # 0/0 in the branch np.where does not take
                W = np.where(ok, new / (sumY / W + Yk / s["W"]), W)
            y1, y2 = Y1 / safe, Yk / safe
            # janafThermo::operator+=
            Tlow = np.where(ok, np.maximum(Tlow, s["Tlow"]), Tlow); Thigh = np.where(ok, np.minimum(Thigh, s["Thigh"]), Thigh)
            high = np.where(ok[:, None], y1[:, None] * high + y2[:, None] * self.high[k], high)
            low = np.where(ok[:, None], y1[:, None] * low + y2[:, None] * self.low[k], low)
            As = np.where(ok, y1 * As + y2 * s["As"], As); Ts = np.where(ok, y1 * Ts + y2 * s["Ts"], Ts)
            sumY = new
        return Mixture(W, Tlow, Thigh, Tcommon, high, low, As, Ts)


def parse_reaction(path):
    """species list and the single irreversible reaction "a A + b B = c C + ..." of the chemistry file"""
    txt = open(path).read()
    species = re.search(r"species\s*\(([^)]*)\)", txt).group(1).split()
    rx = re.search(r'reaction\s+"([^"]*)"', txt).group(1)
    lhs, rhs = rx.split("=")

    def side(s):
        out = []
        for term in s.split("+"):
            mt = re.match(r"\s*([0-9.]*)\s*([A-Za-z][A-Za-z0-9]*)\s*$", term)
            out.append((mt.group(2), float(mt.group(1)) if mt.group(1) else 1.0))
        return out
    return species, side(lhs), side(rhs)


class SingleStep:
    """singleStepReactingMixture: stoichiometry and heat of combustion of the one global reaction"""

    def __init__(self, sp, lhs, rhs, fuel="C3H8", inert="N2"):
        self.sp = sp
        idx = {n: i for i, n in enumerate(sp.names)}
        self.fuelIndex, self.inertIndex = idx[fuel], idx[inert]
        self.O2Index = idx["O2"]
        n = len(sp.names)
        Wu = sp.W[self.fuelIndex]
        hc = np.array([sp.single(i).Hc() for i in range(n)])          # J/kg
        self.stoichCoeffs = np.zeros(n); self.specieProd = np.ones(n)
        q = 0.0
        for name, nu in lhs:
            i = idx[name]
            self.stoichCoeffs[i] = -nu
            q += sp.W[i] * hc[i] * nu / Wu                            # calculateqFuel: += W*hc*stoichCoeff/Wu
        for name, nu in rhs:
            i = idx[name]
            self.stoichCoeffs[i] = nu
            q -= sp.W[i] * hc[i] * nu / Wu
            self.specieProd[i] = -1
        self.qFuel = q
        # massAndAirStoichRatios
        Wm = 0.0
        self.stoicRatio = (sp.W[self.inertIndex] * self.stoichCoeffs[self.inertIndex] + sp.W[self.O2Index] * abs(self.stoichCoeffs[self.O2Index])) \
            / (sp.W[self.fuelIndex] * abs(self.stoichCoeffs[self.fuelIndex]))
        self.s = (sp.W[self.O2Index] * abs(self.stoichCoeffs[self.O2Index])) / (sp.W[self.fuelIndex] * abs(self.stoichCoeffs[self.fuelIndex]))
        # calculateMaxProducts
        prod = [(idx[name], nu) for name, nu in rhs]
        Wm = sum(nu * sp.W[i] for i, nu in prod); tot = sum(nu for _, nu in prod)
        Xprod = {i: nu / tot for i, nu in prod}
        Wm = sum(Xprod[i] * sp.W[i] for i in Xprod)
        self.Yprod0 = np.zeros(n)
        for i in Xprod:
            self.Yprod0[i] = sp.W[i] / Wm * Xprod[i]
        # mass stoichiometric coefficients (specieStoichCoeffs: per kg of fuel)
        self.massCoeffs = self.stoichCoeffs * sp.W / (sp.W[self.fuelIndex] * abs(self.stoichCoeffs[self.fuelIndex]))
