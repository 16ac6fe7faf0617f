const char afx_build_id_str[] = "951c13367211";
