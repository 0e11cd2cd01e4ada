"""Per-element constants used by the pore-geometry path.

Values restate the reference tables (they are inputs to every result):
masses  -> /root/reference/src/pywindow/_internal/tables.py:22-108
vdW radii -> /root/reference/src/pywindow/_internal/tables.py:111-197
covalent radii -> tables.py:200-286 (bond perception in discrete_molecules)
OPLS / DL_F atom-key groups -> tables.py:290-640 (only used by the DL_POLY
ingest to turn force-field keys into elements, once per trajectory).

Layout here is one whitespace table (symbol mass vdw covalent) parsed at import; the
index of a row is the element id handed to the HIP kernels.
"""

from __future__ import annotations

import numpy as np

_ELEMENT_ROWS = """\
AL      26.982 2 1.21
SB      121.76 2 1.39
AR      39.948 1.88 1.51
AS      74.922 1.85 1.21
BA     137.327 2 2.15
BE       9.012 2 0.96
BI      208.98 2 1.48
B       10.811 2 0.83
BR      79.904 1.85 1.21
CD     112.411 1.58 1.54
CS     132.905 2 2.44
CA      40.078 2 1.76
C       12.011 1.7 0.68
CE     140.116 2 2.04
CL      35.453 1.75 0.99
CR      51.996 2 1.39
CO      58.933 2 1.26
CU      63.546 1.4 1.32
DY       162.5 2 1.92
ER      167.26 2 1.89
EU     151.964 2 1.98
F       18.998 1.47 0.64
GD      157.25 2 1.96
GA      69.723 1.87 1.22
GE       72.61 2 1.17
AU     196.967 1.66 1.36
HF      178.49 2 1.75
HE       4.003 1.4 1.5
HO      164.93 2 1.92
H        1.008 1.09 0.23
IN     114.818 1.93 1.42
I      126.904 1.98 1.4
IR     192.217 2 1.41
FE      55.845 2 1.52
KR        83.8 2.02 1.5
LA     138.906 2 2.07
PB       207.2 2.02 1.46
LI       6.941 1.82 1.28
LU     174.967 2 1.87
MG      24.305 1.73 1.41
MN      54.938 2 1.61
HG      200.59 1.55 1.32
MO       95.94 2 1.54
ND      144.24 2 2.01
NE       20.18 1.54 1.5
NI      58.693 1.63 1.24
NB      92.906 2 1.64
N       14.007 1.55 0.68
OS      190.23 2 1.44
O       15.999 1.52 0.68
PD      106.42 1.63 1.39
P       30.974 1.8 1.05
PT     195.078 1.72 1.36
K       39.098 2.75 2.03
PR     140.908 2 2.03
PA     231.036 2 2
RE     186.207 2 1.51
RH     102.906 2 1.42
RB      85.468 2 2.2
RU      101.07 2 1.46
SM      150.36 2 1.98
SC      44.956 2 1.7
SE       78.96 1.9 1.22
SI      28.086 2.1 1.2
AG     107.868 1.72 1.45
NA      22.991 2.27 1.66
SR       87.62 2 1.95
S       32.066 1.8 1.02
TA     180.948 2 1.7
TE       127.6 2.06 1.47
TB     158.925 2 1.94
TL     204.383 1.96 1.45
TH     232.038 2 2.06
TM     168.934 2 1.9
SN      118.71 2.17 1.39
TI      47.867 2 1.6
W       183.84 2 1.62
U      238.029 1.86 1.96
V       50.942 2 1.53
XE      131.29 2.16 1.5
YB      173.04 2 1.87
Y       88.906 2 1.9
ZN       65.39 1.29 1.22
ZR      91.224 2 1.75
X            1 1 1
"""

SYMBOLS: list[str] = []
_mass: list[float] = []
_vdw: list[float] = []
_cov: list[float] = []
for _line in _ELEMENT_ROWS.strip().splitlines():
    _s, _m, _v, _c = _line.split()
    SYMBOLS.append(_s)
    _mass.append(float(_m))
    _vdw.append(float(_v))
    _cov.append(float(_c))

#: element id (row index) by UPPER-CASE symbol
ELEMENT_ID: dict[str, int] = {s: i for i, s in enumerate(SYMBOLS)}
MASS = np.array(_mass, dtype=np.float64)
VDW = np.array(_vdw, dtype=np.float64)
COVALENT = np.array(_cov, dtype=np.float64)
#: elements that end a bond path in discrete_molecules (utilities.py:943)
TERMINAL_SYMBOLS = ("H", "CL", "BR", "F", "HE", "AR", "NE", "KR", "XE", "RN")
#: dict views with the reference's key convention (upper-case symbols)
atomic_mass: dict[str, float] = dict(zip(SYMBOLS, _mass))
atomic_vdw_radius: dict[str, float] = dict(zip(SYMBOLS, _vdw))
atomic_covalent_radius: dict[str, float] = dict(zip(SYMBOLS, _cov))


def element_ids(elements) -> np.ndarray:
    """Map an array of element strings (any case) to int32 element ids.

    Unknown symbols raise ``KeyError`` exactly like the reference's table
    lookup ``atomic_mass[i.upper()]`` (utilities.py:107).
    """
    return np.fromiter(
        (ELEMENT_ID[str(e).upper()] for e in elements),
        dtype=np.int32,
        count=len(elements),
    )


# force-field atom keys -> element.  Keys are case-sensitive: the reference
# lists upper- and lower-case spellings explicitly (tables.py:290-640).
_OPLS_ROWS = """\
Ar: AR Ar ar
B: B b
Br: BR BR- Br br br-
C: CTD CZN C CBO CZB CDS CALK CG CML C5B CTP CTF C5BC CZA CTS CO C5X CQ CP1 CDXR CANI CRA C4T CHZ CAO CTA CDX CA5 CTJ CZ CO4 CTI C5BB CG1 C5M CTM CT C5A CN C3M CB CT1 C5N CO3 CTQ CTH CTU CTE CTC CTG C3T CD CME CT_F CA C56B CT1G C56A CM CTNC CR3 ctd czn c cbo czb cds calk cg cml c5b ctp ctf c5bc cza cts co c5x cq cp1 cdxr cani cra c4t chz cao cta cdx ca5 ctj cz co4 cti c5bb cg1 c5m ctm ct c5a cn c3m cb ct1 c5n co3 ctq cth ctu cte ctc ctg c3t cd cme ct_f ca c56b ct1g c56a cm ctnc cr3
Cl: CL CL- Cl cl cl-
F: F FX1 FX2 FX3 FX4 FG F- f fx1 fx2 fx3 fx4 fg f-
H: HA HAE HS HT3 HC HWS H HNP HAM H_OH HP HT4 HG HMET HO HANI HY HCG HE ha hae hs ht3 hc hws h hnp ham h_oh hp ht4 hg hmet ho hani hy hcg
He: He
I: I I- i i-
Kr: Kr kr
N: NAP NN NB N5BB NS NOM NTC NP N NTH2 NTH NZC NO N5B NO3 NZT NZ NI NTH0 NA5B NT NO2 NBQ NG NE NZA NA NZB NHZ NO2B NEA NA5 nap nn nb n5bb ns nom ntc np n nth2 nth nzc no n5b no3 nzt nz ni nth0 na5b nt no2 nbq ng nza nzb nhz no2b nea na5
Na: Na Na+
Ne: Ne
O: OM OAB ONI O2ZP O2Z OHE OES OBS OT4 OWS O3T OT3 O4T OAL O2 OAS OS ON OVE OZ O OHX OY ONA OA OHP OSP OH om oab oni o2zp o2z ohe oes obs ot4 ows o3t ot3 o4t oal o2 oas os on ove oz o ohx oy ona oa ohp osp oh
P: P P1 P2 P3 P4 PR p p1 p2 p3 p4 pr
Rn: Rn rn
S: S SX6 SY SH SA SZ SD s sx6 sy sh sa sz sd
Xe: Xe xe
"""

OPLS_KEY_TO_ELEMENT: dict[str, str] = {}
for _line in _OPLS_ROWS.strip().splitlines():
    _el, _keys = _line.split(":")
    for _k in _keys.split():
        # first element listing a key wins, as in the reference's search loop
        OPLS_KEY_TO_ELEMENT.setdefault(_k, _el.strip())
