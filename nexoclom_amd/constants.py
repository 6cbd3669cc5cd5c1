"""Physical constants replacing ``astropy.constants`` (astropy 5.3 = CODATA 2018, pinned by the
reference's pyproject.toml:13) and ``periodictable`` 1.6.1 (pyproject.toml:26)."""
G = 6.6743e-11            # m3 kg-1 s-2
H_PLANCK = 6.62607015e-34  # J s
K_B = 1.380649e-23        # J/K
AMU = 1.66053906660e-27   # kg
AU_M = 1.495978707e11     # m
EV = 1.602176634e-19      # J

# periodictable 1.6.1 standard atomic weights for the species the g-value / photo tables hold
ATOMIC_MASS = {
    'H': 1.00794, 'He': 4.002602, 'C': 12.0107, 'N': 14.0067, 'O': 15.9994, 'Na': 22.98976928,
    'Mg': 24.305, 'S': 32.065, 'Cl': 35.453, 'K': 39.0983, 'Ca': 40.078, 'Ti': 47.867,
    'Mn': 54.938045, 'Fe': 55.845, 'Al': 26.9815386, 'Si': 28.0855,
    'OH': 17.00734, 'H2O': 18.01528,
}
