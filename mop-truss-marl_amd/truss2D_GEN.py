"""Drop-in replacement for the reference's truss2D_GEN.py: the parametric 2-row grid truss builder.

`gen_model(num_x, num_y, span_x, span_y, tar_y, dmin, loadx, loady, truss_type='roof',
support_case=1, topo_code=None)` keeps the reference's signature, attributes and side effects
(truss2D_GEN.py:41-115): nodes row-major, element families beams / columns / '\\' braces / '/' braces
(:280-353), supports (:401-418), loads (:421-430), targets (:369-374), first analysis (:433-434).
The analysis itself runs on the MI355X through FEM_2Dtruss.Model.gen_all().  Rendering
(`render_load`) is out of scope (plotting); `savetxt` writes the reference's text format (:193-211).
"""
import csv
import os

import numpy as np

from FEM_2Dtruss import Load, Node, Element, Model
from set_seed_global import seedThis  # noqa: F401  (kept: callers star-import it)

_HERE = os.path.dirname(os.path.abspath(__file__))


def read_section(path):
    """truss2D_GEN.py:29-36; a relative './section_data/...' path falls back to this package's copy."""
    if not os.path.exists(path):
        path = os.path.join(_HERE, path)
    with open(path, newline='') as f:
        data = list(csv.reader(f))
    return np.array(data).astype(float)


class gen_model:
    def __init__(self, num_x, num_y, span_x, span_y, tar_y, dmin, loadx, loady, truss_type='roof', support_case=1,
                 topo_code=None):
        self.model = None
        self._setup(num_x, num_y, span_x, span_y, tar_y, dmin, loadx, loady, truss_type, support_case, topo_code)
        print('------------------------')
        print(self.y_max)
        print(self.y_min)
        print(self.d_min)
        print('------------------------')
        self.gennode()
        self.generate()

    def _setup(self, num_x, num_y, span_x, span_y, tar_y, dmin, loadx, loady, truss_type, support_case, topo_code):
        if num_y != 2:
            raise ValueError("the grid truss has exactly two rows of nodes (num_y == 2)")
        self.num_x, self.num_y = num_x, num_y
        self.span_x, self.span_y = span_x, span_y
        self.tar_y = tar_y
        self.YoungM = 2 * 1e11
        self.truss_path = './section_data/01_brace_rod2.csv'
        self.truss = read_section(self.truss_path)
        self.max_truss_A = self.truss[-1][0] * 1e-4
        self.max_truss_i = self.truss[-1][1] * 1e-8
        self.loadx, self.loady = loadx, loady
        self.truss_type = truss_type
        self.topo_code = topo_code
        self.support_case = support_case
        self.max_poss_brace_vol = 0
        self.max_short_stress = 235 * 1000000
        self.max_long_stress = 235 * 1000000 / 1.5
        self.max_deformation = 0.001 * sum(self.span_x)
        self.y_max = span_y[0]
        self.y_min = 0
        self.d_min = dmin
        self._reset_model()

    def re_value(self, num_x, num_y, span_x, span_y, tar_y, dmin, loadx, loady, truss_type='roof', support_case=1,
                 topo_code=None):
        self._setup(num_x, num_y, span_x, span_y, tar_y, dmin, loadx, loady, truss_type, support_case, topo_code)
        self.gennode()
        self.generate()

    def _reset_model(self):
        self.n_u_x, self.n_u_y, self.n_u_coord, self.n_u_name_div = [], [], [], []
        self.node_bc, self.node_px, self.node_py, self.node_topo, self.node_axial = [], [], [], [], []
        self.E_type1_name, self.E_type2_name, self.E_type3_name, self.E_type4_name = [], [], [], []
        self.model = None

    def set_moveRange(self):
        """truss2D_GEN.py:118-133 (host mirror; the kernel computes the same ranges in float32)."""
        for n in self.model.nodes:
            if n.top_node == 1:
                n.max_up = abs(self.y_max - n.coord[1])
                n.max_down = abs(n.coord[1] - n.vertical_pair[0].coord[1] - self.d_min)
            elif self.truss_type == 'bridge':
                n.max_up = 0
                n.max_down = 0
            elif self.truss_type == 'roof':
                n.max_up = abs(n.vertical_pair[0].coord[1] - n.coord[1] - self.d_min)
                n.max_down = abs(n.coord[1] - self.y_min)

    def gennode(self):
        self.n_u_x = [sum(self.span_x[:i]) for i in range(self.num_x)]
        self.n_u_y = [sum(self.span_y[:i]) for i in range(self.num_y)]
        self.n_u_coord = [[x, y] for y in self.n_u_y for x in self.n_u_x]

    def savetxt(self, name):
        """Structure dump (truss2D_GEN.py:193-211): loads, nodes, elements, one literal per CRLF-ended line.
        Written in numpy 1.x's scalar print mode (the reference pins numpy 1.23.5): under numpy 2.x a
        numpy scalar inside a list prints as `np.float64(..)`, which `read_src` (ast.literal_eval) rejects."""
        import numpy as _np
        try:
            ctx = _np.printoptions(legacy="1.25")
        except (TypeError, ValueError):          # numpy 1.x: already prints plain literals
            import contextlib as _cl
            ctx = _cl.nullcontext()
        with ctx, open(name, "w+") as f:
            for l in self.model.loads:
                f.write(" {}\r\n".format(l))
            for n in self.model.nodes:
                f.write(" {}\r\n".format(n))
            for e in self.model.elements:
                f.write(" {},{},{},{},{},{}\r\n".format(e.name, e.nodes[0].name, e.nodes[1].name, e.em, e.area, e.i))

    def read_src(self, src):
        """Load a structure dump written by `savetxt` back into the model (the reader the reference ships
        with its render scripts, test/*/render/truss2D_READ.py:136-172): one literal per line; 2 items =
        load (name, [Fx, Fy]), 4 = node (name, [x, y], [rx, ry], loads), 6 = element (name, n0, n1, E, A,
        [[I]]); the section number is recovered by matching the area against the section table."""
        import ast
        for line in open(src):
            text = line.rstrip("\r\n").replace(" ", "")
            if not text:
                continue
            val = ast.literal_eval(text)
            if len(val) == 2:
                for l in self.model.loads:
                    if l.name == val[0]:
                        l.size[0], l.size[1] = val[1][0], val[1][1]
            elif len(val) == 4:
                for n in self.model.nodes:
                    if n.name == val[0]:
                        n.coord[0], n.coord[1] = val[1][0], val[1][1]
                        n.res[0], n.res[1] = val[2][0], val[2][1]
            elif len(val) == 6:
                for e in self.model.elements:
                    if e.name == val[0]:
                        e.em, e.area = val[3], val[4]
                        e.i[0][0] = val[5][0][0]
                        for k in range(len(self.truss)):
                            if e.area == self.truss[k][0] * 1e-4:
                                e.section_no = k
            else:
                print('ERROR')
                break

    def _element(self, name, a, b):
        e = Element()
        e.set_name(name)
        e.set_nodes(a, b)
        e.section_no = len(self.truss) - 1
        e.set_em(self.YoungM)
        e.set_area(self.truss[e.section_no][0] * 1e-4)
        e.set_i(self.truss[e.section_no][1] * 1e-8)
        return e

    def generate(self):
        nx = len(self.n_u_x)
        nodes = []
        for i, (x, y) in enumerate(self.n_u_coord):
            n = Node()
            n.set_name(i + 1)
            n.set_coord(x, y)
            nodes.append(n)
        rows = [nodes[r * nx:(r + 1) * nx] for r in range(len(self.n_u_y))]
        self.n_u_name_div = rows
        counter = 1
        self.E_type1_name = []
        for row in rows:                                   # beams, row by row
            for i in range(nx - 1):
                self.E_type1_name.append(self._element(counter, row[i], row[i + 1]))
                counter += 1
        self.E_type2_name = []
        for r in range(len(rows) - 1):                     # columns; they define top_node / vertical_pair
            for i in range(nx):
                lo, hi = rows[r][i], rows[r + 1][i]
                self.E_type2_name.append(self._element(counter, lo, hi))
                if lo.coord[1] > hi.coord[1]:
                    lo.top_node = 1
                else:
                    hi.top_node = 1
                lo.vertical_pair.append(hi)
                hi.vertical_pair.append(lo)
                counter += 1
        self.E_type3_name = []
        for r in range(len(rows) - 1):                     # "\\" braces: upper-left -> lower-right
            for i in range(nx - 1):
                self.E_type3_name.append(self._element(counter, rows[r + 1][i], rows[r][i + 1]))
                counter += 1
        self.E_type4_name = []
        for r in range(len(rows) - 1):                     # "/" braces: lower-left -> upper-right
            for i in range(nx - 1):
                self.E_type4_name.append(self._element(counter, rows[r][i], rows[r + 1][i + 1]))
                counter += 1
        self.model = Model()
        l1 = Load()
        l1.set_name(1)
        l1.set_size(0, self.loady)
        self.model.add_load(l1)
        count_tar = 0
        for n in nodes:
            if n.top_node == 1:
                n.target = self.tar_y[count_tar]
                count_tar += 1
            self.model.add_node(n)
        for fam in (self.E_type1_name, self.E_type2_name, self.E_type3_name, self.E_type4_name):
            for e in fam:
                self.model.add_element(e)
        xs = list(set(n.coord[0] for n in nodes))
        ymin = min(n.coord[1] for n in nodes)
        if self.support_case == 2:
            xs.remove(max(xs))
        elif self.support_case == 3:
            xs.remove(min(xs))
        elif self.support_case == 4:
            xs.remove(min(xs))
            xs.remove(max(xs))
        for n in nodes:
            if n.coord[1] == ymin and (n.coord[0] == max(xs) or n.coord[0] == min(xs)):
                n.set_res(1, 1)
        for n in nodes:
            if self.truss_type == 'bridge':
                loaded = n.coord[1] == ymin and n.res[1] == 0
            else:
                loaded = self.truss_type == 'roof' and n.top_node == 1
            if loaded:
                n.set_load(l1)
                n.has_loady = 1
        self.model.restore()
        self.model.gen_all()
