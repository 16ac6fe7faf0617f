const char afx_build_id_str[] = "f129c2a13365";
