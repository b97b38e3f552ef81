// glue_sanitize.cpp — duckdb_shim/fit_agg_hip.cpp compiled against the DuckDB stand-in (duckdb_stub/) and a MOCK of the
// C ABI (mock_abi.hpp), under -Wall -Wextra with ASan / UBSan: registration (names, aliases, overloads), bind (option
// parsing, result type, Copy / Equals), and the Update / Combine / Finalize / Destroy protocol as a parallel hash
// aggregate, as the naive window aggregator and as a segment tree drive it (glue_driver.hpp).  The mock's "fit" is
// order-sensitive, so a row that reaches the wrong slot, or the right slot in the wrong order, changes the result.
// Test infrastructure only (tests/test_sanitizers_cpu.py); nothing here is shipped.
#include "glue_driver.hpp"

#include "mock_abi.hpp"

#define CHECK(c)                                                           \
	do {                                                                   \
		if (!(c)) {                                                        \
			fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); \
			exit(1);                                                       \
		}                                                                  \
	} while (0)

using namespace glue_driver;

static std::string error_of(const char *fn, const char *spec, bool as_map = false) {
	try {
		Query q(fn, spec, as_map);
	} catch (const std::exception &e) {
		return e.what();
	}
	return "";
}

static void registration_and_bind() {
	Query q("anofox_stats_ols_fit_agg", nullptr, false);
	auto &reg = q.Loader().registered;
	const char *names[3] = {"anofox_stats_ols_fit_agg", "anofox_stats_ridge_fit_agg", "anofox_stats_wls_fit_agg"};
	const char *aliases[3] = {"ols_fit_agg", "ridge_fit_agg", "wls_fit_agg"};
	CHECK(reg.size() == 6);
	for (int k = 0; k < 3; ++k) {
		CHECK(reg.count(names[k]) && reg.count(aliases[k]));
		auto &info = reg.at(names[k]);
		auto &al = reg.at(aliases[k]);
		CHECK(info.alias_of.empty() && al.alias_of == names[k]);
		CHECK(info.on_conflict == OnCreateConflict::ALTER_ON_CONFLICT && al.on_conflict == OnCreateConflict::ALTER_ON_CONFLICT);
		CHECK(info.functions.functions.size() == 2 && al.functions.functions.size() == 2 && info.descriptions.size() == 2);
		const size_t base = k == 2 ? 3 : 2; // wls: (y, x, weight)
		auto &f0 = info.functions.functions[0], &f1 = info.functions.functions[1];
		CHECK(f0.arguments.size() == base && f1.arguments.size() == base + 1);
		CHECK(f0.arguments[0] == LogicalType(LogicalType::DOUBLE) && f0.arguments[1] == LogicalType::LIST(LogicalType::DOUBLE));
		CHECK(f1.arguments.back() == LogicalType(LogicalType::ANY) && f0.return_type == LogicalType(LogicalType::ANY));
		if (k == 2) CHECK(f0.arguments[2] == LogicalType(LogicalType::DOUBLE));
		CHECK(f0.bind && f0.destructor && f0.combine && f0.update && f0.finalize && f0.initialize && !f0.simple_update);
		CHECK(f0.state_size(f0) == 3 * sizeof(int64_t)); // slot, feature count, device shard
	}
	// defaults (ols_aggregate.cpp:48-52): 7 fields, intercept, no inference, 0.95, solver svd, no HC
	CHECK(q.ReturnType().children().size() == 7 && q.ReturnType().children()[0].first == "coefficients" && q.ReturnType().children()[6].first == "n_features");
	{
		const auto o = q.Stats().options;
		CHECK(o.model == ANOFOX_HIP_MODEL_OLS && o.fit_intercept && !o.compute_inference && o.confidence_level == 0.95 && o.solver == ANOFOX_SOLVER_SVD &&
		      o.hc_type == ANOFOX_HC_NONE);
	}
	// the options literal: STRUCT, aliases, case-insensitive keys, integers as booleans, unknown keys ignored
	{
		Query r("ols_fit_agg", "Intercept=0;INFERENCE=true;confidence=0.9;solver=QR;hc_type=HC3;full_output=true", false);
		const auto o = r.Stats().options;
		CHECK(!o.fit_intercept && o.compute_inference && o.confidence_level == 0.9 && o.solver == ANOFOX_SOLVER_QR && o.hc_type == ANOFOX_HC_HC3);
		CHECK(r.ReturnType().children().size() == 14 && r.ReturnType().children()[13].first == "f_pvalue" && r.Inference());
	}
	{ // ridge: alpha wins over lambda, lambda alone counts, scaling; a MAP literal; HC is not a ridge option
		Query r1("anofox_stats_ridge_fit_agg", "lambda=3.0;alpha=0.25;lambda_scaling=glmnet;hc_type=hc1", false);
		CHECK(r1.Stats().options.model == ANOFOX_HIP_MODEL_RIDGE && r1.Stats().options.alpha == 0.25 &&
		      r1.Stats().options.lambda_scaling == ANOFOX_LAMBDA_SCALING_GLMNET && r1.Stats().options.hc_type == ANOFOX_HC_NONE);
		Query r2("ridge_fit_agg", "lambda=3.0;fit_intercept=0.0", true);
		CHECK(r2.Stats().options.alpha == 3.0 && !r2.Stats().options.fit_intercept);
		Query r3("ridge_fit_agg", nullptr, false);
		CHECK(r3.Stats().options.alpha == 1.0 && r3.Stats().options.lambda_scaling == ANOFOX_LAMBDA_SCALING_RAW); // ridge_aggregate.cpp:49-54
		Query r4("anofox_stats_wls_fit_agg", "alpha=5;compute_inference=1", false);
		CHECK(r4.Stats().options.model == ANOFOX_HIP_MODEL_WLS && r4.Stats().options.compute_inference && r4.Stats().options.alpha == 0.0);
	}
	// the reference's error texts (map_options_parser.cpp:21-45,222-266)
	CHECK(error_of("ols_fit_agg", "solver=lu") == "Invalid solver: 'lu'. Valid values are 'qr', 'svd', 'cholesky'");
	CHECK(error_of("ols_fit_agg", "hc_type=hc9") == "Invalid hc_type: 'hc9'. Valid values are 'none', 'hc0', 'hc1', 'hc2', 'hc3'");
	CHECK(error_of("ridge_fit_agg", "lambda_scaling=auto") == "Invalid lambda_scaling: 'auto'. Valid values are 'raw', 'glmnet'");
	CHECK(error_of("ols_fit_agg", "fit_intercept=yes") == "Cannot convert value of type VARCHAR to boolean");
	CHECK(error_of("ols_fit_agg", "fit_intercept=null;solver=svd").empty()); // NULL entries are skipped
	{ // an options argument that does not fold to a constant is not parsed (ols_aggregate.cpp:348): defaults
		Query r("ols_fit_agg", "compute_inference=true;solver=lu", false, /*foldable=*/false);
		CHECK(r.ReturnType().children().size() == 7);
	}
	// Copy() shares the query's arena (every thread must reach the same device state); Equals compares options + arena
	{
		auto c = q.BindData().Copy();
		CHECK(HipAggSharedStateOf(*c) == HipAggSharedStateOf(q.BindData()) && c->Equals(q.BindData()) && q.BindData().Equals(*c));
		Query other("anofox_stats_ols_fit_agg", nullptr, false);
		CHECK(!other.BindData().Equals(q.BindData())); // another aggregate of the query: its own state
	}
}

// rows of the test data and what the mock must return for a list of rows
struct Data {
	size_t n, p;
	std::vector<double> y, x, w;
	std::vector<uint8_t> y_null, x_null, xe_null, w_null;
	std::vector<uint32_t> key;
	Inputs in() const {
		Inputs i;
		i.n = n; i.p = p; i.y = y.data(); i.x = x.data(); i.w = w.data();
		i.y_null = y_null.data(); i.x_null = x_null.data(); i.xe_null = xe_null.data(); i.w_null = w_null.data();
		return i;
	}
};
static Data make_data(size_t n, size_t p, size_t n_keys, unsigned seed, bool nulls) {
	Data d;
	d.n = n; d.p = p;
	unsigned long long rng = seed;
	auto next = [&] { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(rng >> 33); };
	d.y.resize(n); d.x.resize(n * p); d.w.resize(n); d.key.resize(n);
	d.y_null.assign(n, 0); d.x_null.assign(n, 0); d.w_null.assign(n, 0); d.xe_null.assign(n * p, 0);
	for (size_t i = 0; i < n; ++i) {
		d.key[i] = next() % n_keys;
		d.y[i] = (double)(next() % 1000) / 7.0;
		for (size_t j = 0; j < p; ++j) d.x[i * p + j] = (double)(next() % 100) + (double)j;
		d.w[i] = 0.5 + (double)(next() % 4);
		if (nulls) {
			d.y_null[i] = next() % 11 == 0;
			d.x_null[i] = next() % 13 == 0;
			d.w_null[i] = next() % 17 == 0;
		}
	}
	return d;
}

static void group_by(const char *fn, size_t p, int threads, size_t vsize, bool dictionary, int shards = 1) {
	const bool weighted = std::string(fn).find("wls") != std::string::npos;
	const size_t n = 5000, K = 37;
	Data d = make_data(n, p, K, 99 + (unsigned)p, true);
	Query q(fn, nullptr, false);
	Records rec = q.GroupBy(d.in(), d.key.data(), K, threads, vsize, dictionary);
	CHECK(rec.is_null.size() == K);
	// expected order of a key's rows: thread 0's vectors in order, then thread 1's, ... (Combine appends the sources)
	for (size_t k = 0; k < K; ++k) {
		std::vector<Row> rows;
		for (int t = 0; t < threads; ++t) {
			size_t v = 0;
			for (size_t r0 = 0; r0 < n; r0 += vsize, ++v) {
				if ((int)(v % (size_t)threads) != t) continue;
				for (size_t r = r0; r < std::min(n, r0 + vsize); ++r) {
					if (d.key[r] != k || d.y_null[r] || d.x_null[r] || (weighted && d.w_null[r])) continue;
					rows.push_back(Row{d.y[r], d.x[r * p], weighted ? d.w[r] : 1.0});
				}
			}
		}
		std::vector<double> want(p + 6);
		mock_fit(rows, p, want.data());
		if (want[p + 5] != 0.0) { CHECK(rec.is_null[k]); continue; }
		CHECK(!rec.is_null[k] && rec.p == p);
		const double *c = &rec.core[k * (p + 6)];
		for (size_t j = 0; j < p + 5; ++j) CHECK(c[j] == want[j]);
		CHECK(c[p + 5] == (double)p); // n_features
	}
	CHECK(q.Stats().live_slots == 0);                                     // every state was destroyed
	// ONE batched fit for the whole GROUP BY — per device state when ANOFOX_HIP_DEVICES shards the query's states
	CHECK(q.Stats().fit_calls >= 1 && q.Stats().fit_calls <= (uint64_t)shards && q.Stats().slots_fitted >= K - 1);
	if (shards == 1) CHECK(q.Stats().fit_calls == 1);
}

// A prepared statement executed twice: DuckDB re-uses the bind data (and with it the query's device state).  Once the first
// execution's states have all been destroyed, the second one starts from a clean state — slot numbers from 0 again, the same
// results — instead of resting on what the first left behind.
static void prepared_twice() {
	const size_t n = 3000, K = 23, p = 3;
	Data d = make_data(n, p, K, 7, true);
	Query q("ols_fit_agg", nullptr, false);
	const int resets_before = g_reset_calls.load();
	Records r1 = q.GroupBy(d.in(), d.key.data(), K, 3, 128, false);
	const auto slots1 = q.Stats().slot_high_water;
	Records r2 = q.GroupBy(d.in(), d.key.data(), K, 3, 128, false);
	CHECK(g_reset_calls.load() == resets_before + 1);          // exactly one reset: at the second execution's first Initialize
	CHECK(q.Stats().slot_high_water == slots1);                  // the slot numbering started over
	CHECK(r1.is_null == r2.is_null && r1.core == r2.core);      // and the answers are the first execution's
	CHECK(q.Stats().live_slots == 0);
}

static void window_replay() {
	// test/sql/comprehensive_tests.test:425-444: y = 2 i + 1, x = i for i = 1..20, ROWS BETWEEN 4 PRECEDING AND CURRENT ROW
	Data d = make_data(20, 1, 1, 1, false);
	for (size_t i = 0; i < 20; ++i) { d.x[i] = (double)(i + 1); d.y[i] = 2.0 * (double)(i + 1) + 1.0; }
	for (size_t vsize : {(size_t)2048, (size_t)4}) {
		Query q("anofox_stats_ols_fit_agg", nullptr, false);
		Records rec = q.Window(d.in(), 4, vsize);
		CHECK(rec.is_null.size() == 20);
		int n5 = 0;
		for (size_t r = 0; r < 20; ++r) {
			const size_t frame = std::min<size_t>(r + 1, 5);
			if (frame < 2) { CHECK(rec.is_null[r]); continue; } // one row -> NULL (ols_aggregate.cpp:263-267)
			CHECK(!rec.is_null[r] && rec.core[r * 7 + 5] == (double)frame);
			if (rec.core[r * 7 + 5] == 5.0) ++n5;
			double sy = 0;
			for (size_t k = r + 1 - frame; k <= r; ++k) sy += d.y[k];
			CHECK(rec.core[r * 7] == sy); // the mock's first "coefficient": the frame's own rows
		}
		CHECK(n5 == 16);
		CHECK(q.Stats().live_slots == 0 && q.Stats().slots_fitted == 20); // every frame fitted once, nothing re-fitted
		if (vsize == 4) CHECK(q.Stats().slot_high_water <= 8);                 // slots of destroyed states are handed out again
	}
	// a segment tree: leaves of 4 rows, frames of 3 leaves, every leaf the source of up to 3 targets in one Combine call
	{
		Data t = make_data(64, 2, 1, 5, false);
		Query q("ols_fit_agg", nullptr, false);
		Records rec = q.TreeWindow(t.in(), 4, 2, 2048);
		CHECK(rec.is_null.size() == 16);
		for (size_t o = 0; o < 16; ++o) {
			std::vector<Row> rows;
			for (size_t l = o >= 2 ? o - 2 : 0; l <= o; ++l)
				for (size_t r = 4 * l; r < 4 * l + 4; ++r) rows.push_back(Row{t.y[r], t.x[r * 2], 1.0});
			double want[8];
			mock_fit(rows, 2, want);
			CHECK(!rec.is_null[o]);
			for (int j = 0; j < 7; ++j) CHECK(rec.core[o * 8 + j] == want[j]);
		}
		CHECK(q.Stats().live_slots == 0);
	}
}

static void errors_and_flags() {
	{ // LIST lengths that differ (ols_aggregate.cpp:165-175)
		Data d = make_data(10, 3, 1, 3, false);
		std::vector<uint32_t> len(10, 3);
		len[7] = 2;
		Inputs in = d.in();
		in.x_len = len.data();
		Query q("ols_fit_agg", nullptr, false);
		std::string msg;
		try {
			q.GroupBy(in, d.key.data(), 1, 1, 2048, false);
		} catch (const std::exception &e) {
			msg = e.what();
		}
		CHECK(msg == "Inconsistent feature count: expected 3, got 2");
	}
	{ // the feature count is per STATE (:164-175): groups of one query may differ in width, each result has its own LIST lengths
		Data d = make_data(30, 4, 3, 11, false);
		std::vector<uint32_t> len(30);
		for (size_t i = 0; i < 30; ++i) len[i] = d.key[i] == 0 ? 4 : (d.key[i] == 1 ? 2 : 0); // group 2: empty x lists -> NULL
		Inputs in = d.in();
		in.x_len = len.data();
		Query q("ols_fit_agg", "compute_inference=true", false);
		Records rec = q.GroupBy(in, d.key.data(), 3, 2, 7, true);
		CHECK(rec.p == 4 && !rec.is_null[0] && !rec.is_null[1] && rec.is_null[2]);
		CHECK(rec.core[0 * 10 + 4 + 5] == 4.0 && rec.core[1 * 10 + 4 + 5] == 2.0 && std::isnan(rec.core[1 * 10 + 2])); // n_features per row
		size_t n0 = 0, n1 = 0;
		for (size_t i = 0; i < 30; ++i) (d.key[i] == 0 ? n0 : n1) += d.key[i] < 2;
		CHECK(rec.core[0 * 10 + 4 + 4] == (double)n0 && rec.core[1 * 10 + 4 + 4] == (double)n1);
		const auto st = q.Stats();
		CHECK(st.widths == 2 && st.live_slots == 0 && st.rows_accepted == n0 + n1);
		// a later row of another width in the SAME group is the reference's error, with the state's own count
		len[0] = d.key[0] == 0 ? 2 : 4;
		Query q2("ols_fit_agg", nullptr, false);
		std::string msg;
		try {
			q2.GroupBy(in, d.key.data(), 3, 1, 2048, false);
		} catch (const std::exception &e) {
			msg = e.what();
		}
		CHECK(msg.rfind("Inconsistent feature count: expected ", 0) == 0);
		CHECK(q2.Stats().live_slots == 0);
	}
	{ // Combine of states with different feature counts (:217-220): two threads see one group at different widths
		Data d = make_data(8, 3, 1, 12, false);
		std::vector<uint32_t> len = {3, 3, 3, 3, 2, 2, 2, 2};
		Inputs in = d.in();
		in.x_len = len.data();
		Query q("ols_fit_agg", nullptr, false);
		std::string msg;
		try {
			q.GroupBy(in, d.key.data(), 1, 2, 4, false); // vectors of 4 rows, dealt to 2 threads
		} catch (const std::exception &e) {
			msg = e.what();
		}
		CHECK(msg == "Cannot combine states with different feature counts: 3 vs 2" ||
		      msg == "Cannot combine states with different feature counts: 2 vs 3");
	}
	{ // NULL list elements become NaN (the fit's row filter drops such rows): they reach the state as rows
		Data d = make_data(6, 3, 1, 4, false);
		d.xe_null[3 * 3 + 0] = 1;
		Query q("ols_fit_agg", nullptr, false);
		Records rec = q.GroupBy(d.in(), d.key.data(), 1, 1, 2048, true);
		CHECK(!rec.is_null[0] && rec.core[3 + 4] == 6.0 && std::isnan(rec.core[2])); // mock: core[2] = sum x0 w -> NaN
	}
	{ // groups the device state flags as unrefined: SQL NULL, counted, and an error on request
		g_mock_unrefined_rows = 3; // the mock flags every group of exactly 3 rows
		Data d = make_data(9, 1, 3, 8, false);
		for (size_t i = 0; i < 9; ++i) d.key[i] = i < 3 ? 0 : (i < 7 ? 1 : 2); // 3, 4 and 2 rows
		Query q("ols_fit_agg", nullptr, false);
		Records rec = q.GroupBy(d.in(), d.key.data(), 3, 1, 2048, false);
		CHECK(rec.is_null[0] && !rec.is_null[1] && !rec.is_null[2] && q.Stats().unrefined == 1);
		setenv("ANOFOX_HIP_UNREFINED", "error", 1);
		Query q2("ols_fit_agg", nullptr, false);
		std::string msg;
		try {
			q2.GroupBy(d.in(), d.key.data(), 3, 1, 2048, false);
		} catch (const std::exception &e) {
			msg = e.what();
		}
		unsetenv("ANOFOX_HIP_UNREFINED");
		CHECK(msg.find("1 group(s)") != std::string::npos);
		g_mock_unrefined_rows = -1;
	}
}

int main() {
	registration_and_bind();
	group_by("anofox_stats_ols_fit_agg", 3, 1, 2048, false);
	group_by("ols_fit_agg", 3, 4, 64, true);
	group_by("anofox_stats_wls_fit_agg", 8, 3, 100, true);
	group_by("ridge_fit_agg", 12, 5, 2048, false);
	// (r4) the query's states hash-partitioned over three device states: thread-local sources meet targets on other shards in
	// Combine (records exported / imported / merged), every key's rows still arrive once and in order
	setenv("ANOFOX_HIP_DEVICES", "0,0,0", 1);
	{
		const int before = g_export_calls.load();
		group_by("anofox_stats_ols_fit_agg", 3, 4, 64, false, 3);
		group_by("anofox_stats_wls_fit_agg", 8, 3, 100, true, 3);
		CHECK(g_export_calls.load() > before && g_import_calls.load() > 0); // some pairs did cross shards
		group_by("ridge_fit_agg", 12, 5, 2048, false, 1);                   // a log-only width stays on one shard
	}
	unsetenv("ANOFOX_HIP_DEVICES");
	prepared_twice();
	window_replay();
	errors_and_flags();
	CHECK(g_contexts == 0 && g_states == 0 && g_host_allocs == 0); // everything released
	printf("glue_sanitize: all scenarios passed\n");
	return 0;
}
