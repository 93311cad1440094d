// Golden-vector driver: links the UNMODIFIED reference sources (compiled in place from
// /root/reference by oracle/Makefile into oracle/_ref/) and calls their public members with
// fixed inputs.  Nothing in here is shipped; it is test infrastructure that exists only to
// produce the fixtures under tests/golden/ (see oracle/make_golden.py).
//
// usage: ref_driver MODEL data_path K L minCount lambda bias_reg relation_reg alpha
//                   W0.bin|- tuples.txt|- lr out_prefix
//   data_path must look like "01-Data/<name>.txt" relative to the cwd (REBUS::loadPST,
//   REBUS.cpp:8-19, splits on '/' and opens "96-FSUB/<name>_root_fsub_minCount_m_L_l.txt").
//
// Files written (little-endian, no headers; dtypes listed in make_golden.py):
//   <out>.corpus.i32   nUsers nItems nClicks, then per user: n, n*(item,time)   (BEFORE the split)
//   <out>.split.i32    per user: test, test_prev(=val), val, val_prev           (model.hpp:12-53)
//   <out>.meta.txt     NW, num_pos_events, toString()
//   <out>.Winit.f64    W right after init() (rand()-based; only its range is of interest)
//   <out>.W1.f64       W after replaying the tuples through updateFactors
//   <out>.xpos.f64     per user: x_test, x_val (positives' scores, as the model's AUC() computes them)
//   <out>.counts.i32   per user: count_val, count_test, max  (-1 -1 -1 for invalid users)
//   <out>.metrics.f64  AUC_val AUC_test HIT_val HIT_test MRR_val MRR_test var   from md.AUC()
//   <out>.paths.i32    REBUS only: per user, per train position t>=1: len, path...; then per
//                      valid user: len,path (val ctx), len,path (test ctx)
//   <out>.eta.f64      REBUS only: eta_cumWeibullSoftmax rows n=0..L flattened

#include <chrono>
#include "corpus.hpp"
#include "MostPopular.hpp"
#include "BPRMF.hpp"
#include "MC.hpp"
#include "FPMC.hpp"
#include "FossilSimple.hpp"
#include "Fossil.hpp"
#include "HRM_max.hpp"
#include "HRM_avg.hpp"
#include "PRME.hpp"
#include "TransRec.hpp"
#include "TransRec_L1.hpp"
#include "REBUS.hpp"
#include <fstream>

static string g_out;

template <class T>
static void dump(const char* suffix, const vector<T>& v)
{
	string p = g_out + suffix;
	FILE* f = fopen(p.c_str(), "wb");
	if (!f) { fprintf(stderr, "cannot write %s\n", p.c_str()); exit(2); }
	if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), f);
	fclose(f);
}

struct tuple4 { int a, b, c, d; };

static vector<tuple4> read_tuples(const char* path)
{
	vector<tuple4> t;
	if (strcmp(path, "-") == 0) return t;
	FILE* f = fopen(path, "r");
	if (!f) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
	tuple4 x;
	while (fscanf(f, "%d %d %d %d", &x.a, &x.b, &x.c, &x.d) == 4) t.push_back(x);
	fclose(f);
	return t;
}

static void load_W0(model* md, const char* path)
{
	if (strcmp(path, "-") == 0) return;
	FILE* f = fopen(path, "rb");
	if (!f) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
	size_t n = fread(md->W, sizeof(double), md->NW, f);
	if ((int)n != md->NW) { fprintf(stderr, "W0 has %zu values, model wants %d\n", n, md->NW); exit(2); }
	fclose(f);
}

static void dump_W(model* md, const char* suffix)
{
	vector<double> w(md->W, md->W + md->NW);
	dump(suffix, w);
}

// Seed glibc so that the next `rand() % nItems` equals `want` (HRM draws its negative
// inside updateFactors, HRM_avg.cpp:171-174).
static void force_next_rand_mod(int nItems, int want)
{
	for (unsigned s = 1; s < 100000000u; s++) {
		srand(s);
		if (rand() % nItems == want) { srand(s); return; }
	}
	fprintf(stderr, "could not force rand\n");
	exit(2);
}

// Per-user counts with a generic scorer, following model.cpp:16-59.
template <class F>
static void eval_counts(model* md, F score, vector<double>& xpos, vector<int>& counts)
{
	for (int u = 0; u < md->nUsers; u++) {
		int item_test = md->test_per_user[u].first;
		int item_val = md->val_per_user[u].first;
		if (item_test == -1) {
			xpos.push_back(0); xpos.push_back(0);
			counts.push_back(-1); counts.push_back(-1); counts.push_back(-1);
			continue;
		}
		double x_test = score(u, 1, item_test);
		double x_val = score(u, 0, item_val);
		int cv = 0, ct = 0, mx = 0;
		for (int j = 0; j < md->nItems; j++) {
			if (md->clicked_per_user[u].count(j) || j == item_test || j == item_val) continue;
			mx++;
			if (x_val > score(u, 0, j)) cv++;
			if (x_test > score(u, 1, j)) ct++;
		}
		xpos.push_back(x_test); xpos.push_back(x_val);
		counts.push_back(cv); counts.push_back(ct); counts.push_back(mx);
	}
}

static void dump_metrics(model* md)
{
	double m[7];
	md->AUC(&m[0], &m[1], &m[2], &m[3], &m[4], &m[5], &m[6]);
	dump(".metrics.f64", vector<double>(m, m + 7));
}

static void dump_meta(model* md)
{
	string p = g_out + ".meta.txt";
	FILE* f = fopen(p.c_str(), "w");
	fprintf(f, "%d\n%d\n%s\n", md->NW, md->num_pos_events, md->toString().c_str());
	fclose(f);
}

static void dump_split(model* md)
{
	vector<int> s;
	for (int u = 0; u < md->nUsers; u++) {
		s.push_back(md->test_per_user[u].first);
		s.push_back(md->test_per_user[u].second);
		s.push_back(md->val_per_user[u].first);
		s.push_back(md->val_per_user[u].second);
	}
	dump(".split.i32", s);
}

// Simple (user, prev, item) models: everything goes through prediction(u, prev, item).
template <class M>
static void finish_simple(M& md)
{
	vector<double> xpos; vector<int> counts;
	eval_counts(&md, [&](int u, int is_test, int j) {
		int prev = is_test ? md.test_per_user[u].second : md.val_per_user[u].second;
		return md.prediction(u, prev, j);
	}, xpos, counts);
	dump(".xpos.f64", xpos);
	dump(".counts.i32", counts);
	dump_metrics(&md);
}

// "traj:N:C" in place of the tuples file: N epochs of the model's own oneiteration()/oneIteration(),
// the model's own AUC() every C epochs; prints "EPOCH iter seconds" after every epoch and
// "TRAJ iter AUC_val AUC_test HIT_val HIT_test" and "EVALSEC iter seconds" (the time of that AUC() call) after every
// evaluation.
// "sauc:N" in the same place: N calls of the model's own sampleAUC() on the loaded parameters (2000 rand() draws
// per user each, model.cpp:100-167); prints "SAUC call AUC_val AUC_test var".
static int g_traj_n = 0, g_traj_c = 0, g_sauc_n = 0;
#define TRAJ(md, EPOCH_CALL)                                                                  \
	do {                                                                                      \
		for (int sc = 0; sc < g_sauc_n; sc++) {                                               \
			double a_, b_, c_; md.sampleAUC(&a_, &b_, &c_);                                   \
			fprintf(stderr, "SAUC %d %.9f %.9f %.9f\n", sc, a_, b_, c_);                      \
		}                                                                                     \
		if (g_sauc_n) break;                                                                  \
		for (int it = 1; it <= g_traj_n; it++) {                                              \
			const auto t0_ = std::chrono::steady_clock::now();                                \
			md.EPOCH_CALL(lr);                                                                \
			fprintf(stderr, "EPOCH %d %.6f\n", it,                                            \
			        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count()); \
			if (it % g_traj_c == 0) {                                                         \
				const auto e0_ = std::chrono::steady_clock::now();                            \
				double m[7]; md.AUC(&m[0], &m[1], &m[2], &m[3], &m[4], &m[5], &m[6]);         \
				fprintf(stderr, "TRAJ %d %.6f %.6f %.6f %.6f\n", it, m[0], m[1], m[2], m[3]);  \
				fprintf(stderr, "EVALSEC %d %.6f\n", it,                                      \
				        std::chrono::duration<double>(std::chrono::steady_clock::now() - e0_).count()); \
			}                                                                                 \
		}                                                                                     \
	} while (0)

int main(int argc, char** argv)
{
	if (argc != 14) { fprintf(stderr, "bad usage\n"); return 2; }
	if (strncmp(argv[11], "traj:", 5) == 0) { sscanf(argv[11], "traj:%d:%d", &g_traj_n, &g_traj_c); argv[11] = (char*)"-"; }
	if (strncmp(argv[11], "sauc:", 5) == 0) { sscanf(argv[11], "sauc:%d", &g_sauc_n); g_traj_n = 1; g_traj_c = 1; argv[11] = (char*)"-"; }
	string name = argv[1];
	const char* data_path = argv[2];
	int K = atoi(argv[3]);
	int L = atoi(argv[4]);
	int minCount = atoi(argv[5]);
	double lambda = atof(argv[6]);
	double bias_reg = atof(argv[7]);
	double relation_reg = atof(argv[8]);
	double alpha = atof(argv[9]);
	const char* w0 = argv[10];
	vector<tuple4> tuples = read_tuples(argv[11]);
	double lr = atof(argv[12]);
	g_out = argv[13];

	srand(getenv("REF_SEED") ? atoi(getenv("REF_SEED")) : 0);   // main.cpp:144 uses 0
	corpus corp;
	corp.loadData(data_path, 0, 0);
	{
		vector<int> c;
		c.push_back(corp.nUsers); c.push_back(corp.nItems); c.push_back(corp.nClicks);
		for (int u = 0; u < corp.nUsers; u++) {
			c.push_back((int)corp.pos_per_user[u].size());
			for (auto& p : corp.pos_per_user[u]) { c.push_back(p.first); c.push_back(p.second); }
		}
		dump(".corpus.i32", c);
	}

	if (name == "MP") {
		MostPopular md(&corp);
		md.NW = 0; md.W = 0;
		dump_meta(&md); dump_split(&md);
		vector<double> xpos; vector<int> counts;
		// MostPopular::prediction is public; AUC() works through the base class.
		eval_counts(&md, [&](int u, int, int j) { return md.prediction(u, 0, j); }, xpos, counts);
		dump(".xpos.f64", xpos); dump(".counts.i32", counts);
		dump_metrics(&md);
	} else if (name == "BPRMF") {
		BPRMF md(&corp, K, lambda, bias_reg);
		md.init(); dump_meta(&md); dump_split(&md); dump_W(&md, ".Winit.f64"); load_W0(&md, w0);
		if (g_traj_n) { TRAJ(md, oneiteration); return 0; }
		for (auto& t : tuples) md.updateFactors(t.a, t.b, t.c, lr);
		dump_W(&md, ".W1.f64"); finish_simple(md);
	} else if (name == "MC") {
		MC md(&corp, K, lambda);
		md.init(); dump_meta(&md); dump_split(&md); dump_W(&md, ".Winit.f64"); load_W0(&md, w0);
		if (g_traj_n) { TRAJ(md, oneiteration); return 0; }
		for (auto& t : tuples) md.updateFactors(t.a, t.b, t.c, lr);
		dump_W(&md, ".W1.f64"); finish_simple(md);
	} else if (name == "FPMC") {
		FPMC md(&corp, K, K, lambda);
		md.init(); dump_meta(&md); dump_split(&md); dump_W(&md, ".Winit.f64"); load_W0(&md, w0);
		if (g_traj_n) { TRAJ(md, oneiteration); return 0; }
		for (auto& t : tuples) md.updateFactors(t.a, t.b, t.c, t.d, lr);
		dump_W(&md, ".W1.f64"); finish_simple(md);
	} else if (name == "PRME") {
		PRME md(&corp, K, K, alpha, lambda);
		md.init(); dump_meta(&md); dump_split(&md); dump_W(&md, ".Winit.f64"); load_W0(&md, w0);
		if (g_traj_n) { TRAJ(md, oneiteration); return 0; }
		for (auto& t : tuples) md.updateFactors(t.a, t.b, t.c, t.d, lr);
		dump_W(&md, ".W1.f64"); finish_simple(md);
	} else if (name == "HRM_avg") {
		HRM_avg md(&corp, K, lambda);
		md.init(); dump_meta(&md); dump_split(&md); dump_W(&md, ".Winit.f64"); load_W0(&md, w0);
		if (g_traj_n) { TRAJ(md, oneIteration); return 0; }
		for (auto& t : tuples) { force_next_rand_mod(md.nItems, t.d); md.updateFactors(t.a, t.b, t.c, lr); }
		dump_W(&md, ".W1.f64"); finish_simple(md);
	} else if (name == "HRM_max") {
		HRM_max md(&corp, K, lambda);
		md.init(); dump_meta(&md); dump_split(&md); dump_W(&md, ".Winit.f64"); load_W0(&md, w0);
		if (g_traj_n) { TRAJ(md, oneIteration); return 0; }
		for (auto& t : tuples) { force_next_rand_mod(md.nItems, t.d); md.updateFactors(t.a, t.b, t.c, lr); }
		dump_W(&md, ".W1.f64"); finish_simple(md);
	} else if (name == "TransRec") {
		TransRec md(&corp, K, lambda, relation_reg, bias_reg);
		md.init(); dump_meta(&md); dump_split(&md); dump_W(&md, ".Winit.f64"); load_W0(&md, w0);
		if (g_traj_n) { TRAJ(md, oneIteration); return 0; }
		for (auto& t : tuples) md.updateFactors(t.a, t.b, t.c, t.d, lr);
		dump_W(&md, ".W1.f64"); finish_simple(md);
	} else if (name == "TransRec_L1") {
		TransRec_L1 md(&corp, K, lambda, relation_reg, bias_reg);
		md.init(); dump_meta(&md); dump_split(&md); dump_W(&md, ".Winit.f64"); load_W0(&md, w0);
		if (g_traj_n) { TRAJ(md, oneIteration); return 0; }
		for (auto& t : tuples) md.updateFactors(t.a, t.b, t.c, t.d, lr);
		dump_W(&md, ".W1.f64"); finish_simple(md);
	} else if (name == "FossilSimple") {
		FossilSimple md(&corp, K, lambda, bias_reg);
		md.init(); dump_meta(&md); dump_split(&md); dump_W(&md, ".Winit.f64"); load_W0(&md, w0);
		if (g_traj_n) { TRAJ(md, oneiteration); return 0; }
		for (auto& t : tuples) md.updateFactors(t.a, t.b, t.c, t.d, lr);
		dump_W(&md, ".W1.f64"); finish_simple(md);
	} else if (name == "Fossil") {
		Fossil md(&corp, L, K, lambda, bias_reg);
		md.init(); dump_meta(&md); dump_split(&md); dump_W(&md, ".Winit.f64"); load_W0(&md, w0);
		if (g_traj_n) { TRAJ(md, oneiteration); return 0; }
		// tuple = (user, target position t>=1 in the train sequence, neg, unused);
		// prev_items as Fossil::oneiteration builds them (Fossil.cpp:266-272).
		for (auto& t : tuples) {
			vector<pair<int,int> >& up = corp.pos_per_user[t.a];
			vector<int> prev;
			for (int j = t.b - 1; j >= max(0, t.b - L); j--) prev.push_back(up[j].first);
			md.updateFactors(t.a, prev, up[t.b].first, t.c, lr);
		}
		dump_W(&md, ".W1.f64");
		vector<double> xpos; vector<int> counts;
		eval_counts(&md, [&](int u, int is_test, int j) {
			// contexts as Fossil::AUC builds them (Fossil.cpp:387-404)
			vector<pair<int,int> >& up = corp.pos_per_user[u];
			vector<int> prev;
			if (is_test) {
				prev.push_back(md.test_per_user[u].second);
				for (int i = 1; i <= L - 1; i++) { if ((int)up.size() - i < 0) break; prev.push_back(up[up.size() - i].first); }
			} else {
				for (int i = 1; i <= L; i++) { if ((int)up.size() - i < 0) break; prev.push_back(up[up.size() - i].first); }
			}
			return md.prediction(u, prev, j);
		}, xpos, counts);
		dump(".xpos.f64", xpos); dump(".counts.i32", counts);
		dump_metrics(&md);
	} else if (name == "REBUS") {
		REBUS md(&corp, K, lambda, bias_reg, alpha);
		if (md.loadPST(minCount, L, data_path) != 0) { fprintf(stderr, "no FSUB file\n"); return 3; }
		md.init(); dump_meta(&md); dump_split(&md); dump_W(&md, ".Winit.f64"); load_W0(&md, w0);
		if (g_traj_n) { TRAJ(md, oneiteration); return 0; }
		{
			vector<double> eta;
			for (auto& row : md.eta_cumWeibullSoftmax) for (double e : row) eta.push_back(e);
			dump(".eta.f64", eta);
		}
		// tuple = (user, index into matrix[u] = target position - 1, neg, unused), REBUS.cpp:540-554
		for (auto& t : tuples) {
			auto& it = md.matrix[t.a].at(t.b);
			vector<int> prev = it.second.first;
			unordered_set<int> prev_set = it.second.second;
			vector<int> path = md.histo_user[t.a].at(t.b).second;
			md.updateFactors(t.a, path, prev, prev_set, it.first, t.c, lr);
		}
		dump_W(&md, ".W1.f64");
		vector<int> paths;
		for (int u = 0; u < md.nUsers; u++)
			for (auto& h : md.histo_user[u]) { paths.push_back((int)h.second.size()); for (int x : h.second) paths.push_back(x); }
		// eval contexts, REBUS.cpp:758-782
		vector<vector<int> > vp(md.nUsers), tp(md.nUsers), vpath(md.nUsers), tpath(md.nUsers);
		vector<unordered_set<int> > vs(md.nUsers), ts(md.nUsers);
		for (int u = 0; u < md.nUsers; u++) {
			if (md.test_per_user[u].first == -1) continue;
			for (auto& p : corp.pos_per_user[u]) { vp[u].push_back(p.first); tp[u].push_back(p.first); }
			tp[u].push_back(md.test_per_user[u].second);
			vs[u] = unordered_set<int>(vp[u].begin(), vp[u].end());
			ts[u] = unordered_set<int>(tp[u].begin(), tp[u].end());
			vpath[u] = md.findPathStars(vp[u]);
			tpath[u] = md.findPathStars(tp[u]);
			paths.push_back((int)vpath[u].size()); for (int x : vpath[u]) paths.push_back(x);
			paths.push_back((int)tpath[u].size()); for (int x : tpath[u]) paths.push_back(x);
		}
		dump(".paths.i32", paths);
		vector<double> xpos; vector<int> counts;
		eval_counts(&md, [&](int u, int is_test, int j) {
			return is_test ? md.prediction(u, tpath[u], tp[u], ts[u], j)
			               : md.prediction(u, vpath[u], vp[u], vs[u], j);
		}, xpos, counts);
		dump(".xpos.f64", xpos); dump(".counts.i32", counts);
		dump_metrics(&md);
	} else {
		fprintf(stderr, "unknown model %s\n", name.c_str());
		return 2;
	}
	return 0;
}
